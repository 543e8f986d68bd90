"""The documents cite measurement records by path; a record that is cited must exist (a reviewer must be able to find
the current state - and every number - without archaeology), and the profile set of the sources in the tree must be the committed one."""
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOCS = ("DESIGN.md", "README.md", "INTEGRATION.md", "tools/README.md", "oracle/README.md")


def _cited_paths(text):
    out = set()
    for m in re.finditer(r"`((?:profiles|tests/golden|tools|oracle|include|actionmesh_amd)/[A-Za-z0-9_./{},*-]+)`", text):
        p = m.group(1).rstrip(".,")
        if "{" in p:                                   # brace lists: profiles/r05c_energy_table.{txt,json}
            head, rest = p.split("{", 1)
            alts, tail = rest.split("}", 1)
            out.update(head + a + tail for a in alts.split(","))
        else:
            out.add(p)
    return out


def test_every_cited_record_exists():
    missing = []
    for doc in DOCS:
        with open(os.path.join(ROOT, doc)) as f:
            text = f.read()
        for p in sorted(_cited_paths(text)):
            if p.endswith("/") or "<" in p:
                continue
            if not glob.glob(os.path.join(ROOT, p)) and not glob.glob(os.path.join(ROOT, p + "*")):
                missing.append(f"{doc}: {p}")
    assert not missing, "cited but absent:\n" + "\n".join(missing)


def test_design_is_a_current_state_document():
    with open(os.path.join(ROOT, "DESIGN.md")) as f:
        lines = f.read().splitlines()
    assert len(lines) <= 400, f"DESIGN.md has {len(lines)} lines; the lab notebook belongs in profiles/HISTORY.md"
    assert os.path.exists(os.path.join(ROOT, "profiles", "HISTORY.md"))


def test_committed_profile_set_matches_the_sources():
    """bench.py quotes roofline.traffic only from a record that names the sha of the kernel sources in the tree: the round must end with
    that record (and the kernel trace / PMC summaries of the same sha) committed."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    sha = bench.source_sha()
    recs = glob.glob(os.path.join(ROOT, "profiles", f"*_{sha}_attention_traffic.json"))
    assert recs, f"no profiles/*_{sha}_attention_traffic.json for the sources in the tree"
    with open(recs[0]) as f:
        rec = json.load(f)
    assert rec["source_sha"] == sha and rec["traffic_bytes_per_launch"] > rec["algorithmic_unique_bytes_per_launch"] > 0
    for kind in ("kernel_stats.csv", "pmc.csv", "dominant_kernel.csv", "bench.json"):
        assert glob.glob(os.path.join(ROOT, "profiles", f"*_{sha}_{kind}")), kind
