"""Derived busy shares from a tools/pmc_kernels.sh summary: prepends the table profiles/rNN_pmc_step.md opens with.

    python3 tools/pmc_derive.py <raw summary .md> <out .md> "<title>"
"""
import re
import sys

raw, out, title = sys.argv[1:4]
text = open(raw).read()
rows = []
for block in re.split(r"\n## ", text)[1:]:
    name = block.split("\n", 1)[0].strip().strip("`")
    vals = {m.group(1): float(m.group(2)) for m in re.finditer(r"\| (\w+) \| ([0-9.e+\-]+) \|", block)}
    need = ("GRBM_GUI_ACTIVE", "SQ_ACTIVE_INST_VALU", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES")
    if not all(k in vals for k in need) or vals["GRBM_GUI_ACTIVE"] <= 0:
        continue
    cyc = vals["GRBM_GUI_ACTIVE"] / 8.0  # the counter sums the 8 XCDs
    rows.append((name, cyc / 1e6, 100.0 * vals["SQ_ACTIVE_INST_VALU"] * 4.0 / (cyc * 1024.0),
                 100.0 * vals["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0),
                 100.0 * vals["SQ_WAIT_INST_ANY"] / max(vals["SQ_WAVE_CYCLES"], 1.0)))
head = [f"# {title}\n",
        "Derived per kernel (per dispatch): GPU cycles = GRBM_GUI_ACTIVE / 8 XCDs; vector ALU busy = SQ_ACTIVE_INST_VALU x 4 /",
        "(cycles x 1024 SIMDs); matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024); waiting = SQ_WAIT_INST_ANY /",
        "SQ_WAVE_CYCLES.  The f32 matrix instructions and the vector ALU share a SIMD's issue: their shares add.",
        "(tools/pmc_kernels.sh + tools/pmc_derive.py)\n",
        "| kernel | M cycles | vector ALU busy % | matrix pipe busy % | wave cycles waiting for an instruction % |",
        "|---|---|---|---|---|"]
for r in sorted(rows):
    head.append(f"| `{r[0]}` | {r[1]:.2f} | {r[2]:.1f} | {r[3]:.1f} | {r[4]:.1f} |")
open(out, "w").write("\n".join(head) + "\n\n" + text)
print(f"{len(rows)} kernels")
