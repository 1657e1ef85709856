"""The committed evidence is self-consistent (CPU, no GPU): every bench line and PMC file of the current round under profiles/ names ONE
binary — the one whose hash is in profiles/<round>_libhj.sha256 — and the result tables in README / DESIGN / BASELINE are what
tools/update_docs.py writes from those files (a table edited by hand, or evidence refreshed without the documents, fails here)."""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = "r6"


def _lines(path):
    with open(path) as f:
        text = f.read()
    try:
        return [json.loads(text)]
    except json.JSONDecodeError:   # several bench lines in one file
        return [json.loads(l) for l in text.splitlines() if l.startswith("{")]


def test_every_bench_line_and_pmc_file_names_the_same_binary():
    want = open(os.path.join(ROOT, "profiles", ROUND + "_libhj.sha256")).read().split()[0]
    assert len(want) == 64
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", ROUND + "_bench_*.json")) + glob.glob(os.path.join(ROOT, "profiles", ROUND + "_pmc_*.json")))
    assert len(files) >= 20
    seen = 0
    for f in files:
        for d in _lines(f):
            if "lib_sha256" in d:
                assert d["lib_sha256"] == want, (os.path.basename(f), d["lib_sha256"][:12], want[:12])
                seen += 1
    assert seen >= 20
    # the headline-class lines carry HBM traffic from the PMC files of that same binary (bench.py looks the hash up before it quotes them)
    for name in ("2p30", "2p27", "zipf"):
        d = _lines(os.path.join(ROOT, "profiles", "%s_bench_%s.json" % (ROUND, name)))[0]
        r = d["roofline"]
        assert r["traffic"] and r["achieved"] and 0.3 < r["frac"] < 1.0, (name, r)
        assert d["cpu_baseline"]["value"] > 0


def test_generated_tables_match_the_profiles():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "update_docs.py"), "--check"], capture_output=True, text=True,
                       env=dict(os.environ, ROUND=ROUND), cwd=ROOT)
    assert p.returncode == 0 and "stale: nothing" in p.stdout, p.stdout + p.stderr
