"""Per-kernel counter summary of the passes tools/pmc_kernels.sh collected: <dir> <kernel regex> <out.md> <command...>"""
import collections, csv, glob, os, re, sys
d, filt, out = sys.argv[1:4]
cmd = " ".join(sys.argv[4:])
csv.field_size_limit(1 << 30)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(lambda: collections.defaultdict(set))
meta = {}
for path in sorted(glob.glob(os.path.join(d, "g*", "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(path)):
        m = re.search(r"(\w+_kernel(?:<[^>(]*>)?)", r["Kernel_Name"])
        k = m.group(1) if m else r["Kernel_Name"][:40]
        if not re.search(filt, k):
            continue
        c = r["Counter_Name"]
        agg[k][c] += float(r["Counter_Value"])
        disp[k][c].add(r["Dispatch_Id"])
        meta[k] = (r["Grid_Size"], r["Workgroup_Size"], r["LDS_Block_Size"], r["VGPR_Count"], r["Accum_VGPR_Count"])
lines = [f"# Hardware counters per kernel\n", f"command: `{cmd}`  (one `rocprofv3 --pmc` pass per counter group, values per dispatch)\n"]
failed = os.path.join(d, "failed.txt")
if os.path.exists(failed):
    lines.append("passes that did not run: " + "; ".join(l.strip() for l in open(failed)) + "\n")
for k in sorted(agg):
    g = meta[k]
    lines.append(f"\n## `{k}`\ngrid {g[0]} threads, workgroup {g[1]}, LDS {g[2]} B, VGPRs {g[3]} + {g[4]} accumulation\n")
    lines.append("| counter | per dispatch | dispatches |\n|---|---|---|")
    v = {}
    for c in sorted(agg[k]):
        n = len(disp[k][c])
        v[c] = agg[k][c] / n
        lines.append(f"| {c} | {v[c]:.6g} | {n} |")
    der = []
    if "SQ_WAVE_CYCLES" in v and v["SQ_WAVE_CYCLES"]:
        for c, what in (("SQ_WAIT_INST_ANY", "wave cycles waiting for an instruction's operands / issue"), ("SQ_WAIT_ANY", "wave cycles waiting on anything"),
                        ("SQ_ACTIVE_INST_ANY", "wave cycles with an instruction in flight")):
            if c in v:
                der.append(f"{what}: {100 * v[c] / v['SQ_WAVE_CYCLES']:.1f} % of SQ_WAVE_CYCLES")
    if "SQ_BUSY_CYCLES" in v and v["SQ_BUSY_CYCLES"]:
        for c in ("SQ_ACTIVE_INST_VALU", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT"):
            if c in v:
                der.append(f"{c} / SQ_BUSY_CYCLES = {v[c] / v['SQ_BUSY_CYCLES']:.3f}")
    if "SQ_WAVES" in v and v["SQ_WAVES"]:
        for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM"):
            if c in v:
                der.append(f"{c} per wave = {v[c] / v['SQ_WAVES']:.1f}")
    if "TCC_HIT_sum" in v and "TCC_MISS_sum" in v and v["TCC_HIT_sum"] + v["TCC_MISS_sum"]:
        der.append(f"L2 hit rate {100 * v['TCC_HIT_sum'] / (v['TCC_HIT_sum'] + v['TCC_MISS_sum']):.1f} %; TCC_MISS x 64 B = {v['TCC_MISS_sum'] * 64 / 1e9:.3f} GB")
    if "FETCH_SIZE" in v:
        der.append(f"FETCH_SIZE = {v['FETCH_SIZE'] * 1024 / 1e9:.3f} GB raw (KB units) = {2 * v['FETCH_SIZE'] * 1024 / 1e9:.3f} GB of memory-side reads "
                   "(gfx950 tallies a 128-byte request as 64; the requests by size of profiles/r05_pmc_traffic_* show every kernel's reads are 128-byte requests)")
    if "WRITE_SIZE" in v:
        der.append(f"WRITE_SIZE = {v['WRITE_SIZE'] * 1024 / 1e9:.3f} GB")
    if "TCP_PENDING_STALL_CYCLES_sum" in v and "GRBM_GUI_ACTIVE" in v and v["GRBM_GUI_ACTIVE"]:
        der.append(f"TCP_PENDING_STALL_CYCLES_sum / (GRBM_GUI_ACTIVE x 256 CUs) = {v['TCP_PENDING_STALL_CYCLES_sum'] / v['GRBM_GUI_ACTIVE'] / 256:.3f}")
    if der:
        lines.append("\nderived: " + "; ".join(der))
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:60]))
