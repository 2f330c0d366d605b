"""A/B of environment switches on the headline step, in one gpurun call (development helper).

    python3 tools/ab_env.py [--workload W] [--rounds R] "NAME=VAL ..." "NAME=VAL ..." ...

Every argument is one variant: a space-separated list of environment assignments ("" or "-" = the default build).
Each variant runs `bench.py --steps 5 --warmup 3` (no CPU / front-end / secondary legs) in its own child process,
the variants alternating for R rounds; prints ms/step and the per-class ms of every run and the medians."""
import json, os, statistics, subprocess, sys

args = sys.argv[1:]
workload, rounds = "coif4-l14", 2
while args and args[0].startswith("--"):
    if args[0] == "--workload":
        workload = args[1]
    elif args[0] == "--rounds":
        rounds = int(args[1])
    args = args[2:]
variants = args or ["-"]
res = {v: [] for v in variants}
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for r in range(rounds):
    for v in variants:
        env = dict(os.environ)
        for kv in v.split():
            if "=" in kv:
                k, val = kv.split("=", 1)
                env[k] = val
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", workload, "--steps", "5", "--warmup", "3",
                              "--cpu-frames", "0", "--e2e-steps", "0", "--no-frontends", "--no-secondary"],
                             env=env, capture_output=True, text=True, cwd=root)
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not lines:
            print(v, "FAILED", out.stderr[-800:], flush=True)
            continue
        d = json.loads(lines[-1])
        cls = {k: round(c["ms_per_step"], 3) for k, c in d["classes"].items()}
        res[v].append((d["ms_per_step"], cls))
        print(f"round {r} [{v}] {d['ms_per_step']:.3f} ms/step  {cls}", flush=True)
print()
for v, runs in res.items():
    if not runs:
        continue
    med = statistics.median(x[0] for x in runs)
    keys = runs[0][1].keys()
    print(f"[{v}] median {med:.3f} ms/step  " + "  ".join(f"{k} {statistics.median(x[1][k] for x in runs):.3f}" for k in keys))
