"""Random models and calls through the fused LBS kernels (smil_lbs_forward_project, smil_lbs_backward_ndc) against the separate
kernels AND against the CPU oracle's autograd: tools/dbg/fuzz_lbs.py <seed0> <seed1> [wide].  The cases themselves live in
tests/lbs_cases.py (a dozen of them run under pytest -m gpu, tests/test_gpu_lbs_fused.py)."""
import os
import sys

REPO = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import lbs_cases  # noqa: E402

WIDE = len(sys.argv) > 3 and sys.argv[3] == "wide"  # up to 250 joints and 20 views
bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    checks, i = lbs_cases.run_case(seed, WIDE)
    fails = [(w, e) for f, w, e in checks if f is not None]
    worst = max(e for _, _, e in checks)
    print(f"seed {seed:3d} V={i['V']:4d} J={i['J']:3d} nB={i['nB']} static={i['static']} B={i['B']:2d} views={i['views']} shared_beta={i['shared_beta']} "
          f"trans_after={i['trans_after']} fused_bwd={i['fused_bwd']} worst rel {worst:.1e} {'ok' if not fails else 'FAIL ' + str(fails)}", flush=True)
    bad += bool(fails)
print("failures", bad)
