"""Summarise gpurun_out/models (tools/profile_models.sh) into profiles/rNN_<model>_kernel_stats.txt."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_profiles import ROOT, kernel_stats  # noqa: E402


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    src = os.path.join(ROOT, "gpurun_out", "models")
    for w in ("tgat", "tisasrec", "ctsma"):
        db = os.path.join(src, w, "k_results.db")
        if not os.path.exists(db):
            continue
        steps = 15
        head = [f"# rocprofv3 --kernel-trace --stats -- python bench.py --workload {w} --steps 10 --warmup 5   ({steps} optimizer steps, autograd path, round {tag})"]
        bl = os.path.join(src, w + ".json")
        if os.path.exists(bl):
            head.append("# bench line of the same (profiled) run: " + open(bl).read().strip())
        with open(os.path.join(ROOT, "profiles", f"{tag}_{w}_kernel_stats.txt"), "w") as f:
            f.write("\n".join(head + kernel_stats(db, steps)[:40]) + "\n")
        print("written", w)


if __name__ == "__main__":
    main()
