import sys, os
REPO = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import lbs_cases
for seed in (12025,):
    checks, info = lbs_cases.run_case(seed)
    print(info)
    for f, w, e in checks:
        print(f"  {w:22s} {e:.3e} {'FAIL' if f is not None else ''}")
