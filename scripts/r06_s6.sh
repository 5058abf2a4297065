#!/bin/bash
# round 6 session 6: counters (separate --pmc passes, profiles/run_profiles.sh) behind the A/B of session 5 — the tiled kernel at C2
# as shipped, with the dense record grid (cand_dense), with the particles ordered by pose (--sort-poses cluster)
O=gpurun_out/r06g; mkdir -p $O
bash profiles/run_profiles.sh r06g_C2base --workload C2 > $O/prof_base.log 2>&1
MCL3DL_HIP_OPTIONS="cand_dense=1,index_budget_bytes=0" bash profiles/run_profiles.sh r06g_C2dense --workload C2 > $O/prof_dense.log 2>&1
bash profiles/run_profiles.sh r06g_C2cluster --workload C2 --sort-poses cluster > $O/prof_cluster.log 2>&1
for t in base dense cluster; do cp gpurun_out/prof_r06g_C2$t/r06g_C2${t}_kernel_stats.csv gpurun_out/prof_r06g_C2$t/r06g_C2${t}_pmc_summary.csv $O/ 2>/dev/null; grep "likelihood_tiled" $O/r06g_C2${t}_pmc_summary.csv | grep -E "SQ_INSTS_VALU,|TCP_TCC_READ_REQ|TCC_MISS|TCC_HIT|SQ_INSTS_VMEM_RD|GRBM_GUI|FETCH_SIZE|SQ_WAVE_CYCLES|SQ_WAIT_INST_ANY|SQ_BUSY" | sed "s/^/$t: /"; done
