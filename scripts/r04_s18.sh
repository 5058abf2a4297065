#!/bin/bash
# round-4 PMC sessions: counters tied to the sources of commit ba862c69c79d (profiles/summarize_pmc.py writes the source shas)
export GIT_HEAD=ba862c69c79d
bash profiles/run_profiles.sh r04z_C2 --workload C2 2>&1 | tail -12
bash profiles/run_profiles.sh r04z_C2j --workload C2 --map-jitter 0.045 2>&1 | tail -12
bash profiles/run_profiles.sh r04z_C3 --workload C3 2>&1 | tail -12
