"""Register / scratch / LDS use of every kernel of one HIP source, from hipcc's resource-usage remarks (no GPU).

    python3 tools/kres.py audiodeepfake-detection_amd/csrc/wino44.hip [substring filter]
"""
import re, subprocess, sys

src = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950",
                      "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"],
                     capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"remark: (?:\S+ )?Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    m = re.search(r"remark: \s*(\w[\w ]*?): (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
dem = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
for r, d in zip(rows, dem):
    d = re.sub(r"\(anonymous namespace\)::", "", d)
    d = re.sub(r"\(.*", "", d)
    if filt and filt not in d:
        continue
    print(f"{d[:78]:78s} sgpr {r.get('TotalSGPRs', r.get('SGPRs', 0)):4d} vgpr {r.get('VGPRs', 0):4d} agpr {r.get('AGPRs', 0):4d} "
          f"scratch {r.get('ScratchSize [bytes/lane]', 0):5d} occ {r.get('Occupancy [waves/SIMD]', 0)} lds {r.get('LDS Size [bytes/block]', 0)}")
