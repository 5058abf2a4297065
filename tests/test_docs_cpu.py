"""Every file the documents cite under profiles/, scripts/, tests/ or the source tree exists (DESIGN.md once cited a profile
summary that was never committed)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOCS = ["DESIGN.md", "README.md", "INTEGRATION.md", "profiles/README.md", "scripts/README.md"]
PAT = re.compile(r"`((?:profiles|scripts|tests|oracle|include|mcl_3dl_amd)/[A-Za-z0-9_./{},*-]+)`")
BARE = re.compile(r"`(r0\d[a-z]?\d?_[A-Za-z0-9_.{},*-]+\.(?:json|csv|txt))`")   # a profile cited by its file name alone


def expand(path):
    """`a/{x,y}_z.csv` -> both; a trailing wildcard or a directory is checked as a prefix."""
    m = re.search(r"\{([^{}]*)\}", path)
    if not m:
        return [path]
    out = []
    for alt in m.group(1).split(","):
        out += expand(path[:m.start()] + alt + path[m.end():])
    return out


def test_cited_files_exist():
    missing = []
    for doc in DOCS:
        text = open(os.path.join(ROOT, doc)).read()
        for cite in PAT.findall(text):
            for p in expand(cite.split("::")[0].rstrip(".,")):
                p = p.split(":")[0]
                full = os.path.join(ROOT, p)
                if "*" in p:
                    import glob
                    ok = bool(glob.glob(full))
                else:
                    ok = os.path.exists(full) or p.endswith(".so") or "/_ref/" in p or p.endswith(".bin")   # built artefacts
                if not ok:
                    missing.append((doc, cite))
        for cite in BARE.findall(text):
            import glob
            for p in expand(cite):
                if not glob.glob(os.path.join(ROOT, "profiles", p)):
                    missing.append((doc, cite))
    assert not missing, missing


def test_the_patterns_find_something():
    # (DESIGN.md = the current design, ~25 KB; HISTORY.md = rounds 1-4 measurement by measurement: it cites session drivers that
    # live in the git history only and is therefore not in DOCS — every profiles/ file it names is still checked here)
    text = open(os.path.join(ROOT, "DESIGN.md")).read()
    assert len(PAT.findall(text)) > 30
    hist = open(os.path.join(ROOT, "HISTORY.md")).read()
    import glob
    assert len(BARE.findall(hist)) > 20
    missing = [c for c in BARE.findall(hist) for p in expand(c) if not glob.glob(os.path.join(ROOT, "profiles", p))]
    assert not missing, missing
