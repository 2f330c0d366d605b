"""Instruction-level floor of the level-14 front end (task 4: "or profiles/r05_frontend_floor.md with per-stage
instruction counts x issue cycles showing the floor and what fraction of 0.60 it allows").  No GPU: arithmetic on the
kernels' geometry + the per-wave instruction counts of profiles/r05_pmc_frontend_{coif4,sym5}.md.

    python3 tools/frontend_floor.py > profiles/r05_frontend_floor.md
"""
import re

PEAK = 8000.0  # GB/s
CUS, SIMDS = 256, 4
GHZ = 2.0      # shader clock the chip holds under these kernels (rocprof time / GRBM_GUI_ACTIVE cycles of the same kernels)
B = 4096


def node_lengths(L, levels=14, n=22050):
    out = []
    for _ in range(levels):
        n = (n + L - 2 + (n & 1)) // 2
        out.append(n)
    return out


def counts(path):
    txt = open(path).read()
    res = {}
    for name, body in re.findall(r"## `(\w+)<[^`]*`\n(.*?)(?=\n## |\Z)", txt, re.S):
        g = lambda k: float(re.search(k + r" per wave = ([\d.]+)", body).group(1))
        cyc = float(re.search(r"\| GRBM_GUI_ACTIVE \| ([\d.e+]+)", body).group(1)) / 8
        act = float(re.search(r"SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES = ([\d.]+)", body).group(1)) * float(re.search(r"\| SQ_BUSY_CYCLES \| ([\d.e+]+)", body).group(1))
        res[name] = {"valu": g("SQ_INSTS_VALU"), "salu": g("SQ_INSTS_SALU"), "lds": g("SQ_INSTS_LDS"), "vmem": g("SQ_INSTS_VMEM"),
                     "cycles": cyc, "valu_busy": act * 4 / (cyc * CUS * SIMDS)}
    return res


def analyse(name, L, pmc_path, measured_us, feat_bytes):
    K = L // 2
    n = node_lengths(L)
    c = counts(pmc_path)
    top, deep = c["wpt3_top_kernel"], c["wpt4_deep_kernel"]
    # top kernel, direct form, workgroup = one level-1 half (1024 threads): packed FMAs (two multiply-adds each)
    pk_top = n[0] * L // 2 + sum((1 << (k - 2)) * n[k - 1] * L for k in range(2, 9))
    pos_top = n[0] + sum((1 << (k - 2)) * n[k - 1] for k in range(2, 9))
    fma_top = pk_top / 1024
    # deep kernel, lattice form, workgroup = 16 level-8 nodes (512 threads): K packed FMAs per position, K - 1 positions of
    # lead-in per node; epilogue: one quarter-rate v_log_f32 per coefficient
    pk_deep = sum(16 * (1 << (k - 9)) * (n[k - 1] + K - 1) * K for k in range(9, 15))
    fma_deep = pk_deep / 512
    logs = 1024 * n[13] / 512
    print(f"\n## {name} (L = {L}, K = {K}; node lengths {n[:8]} | {n[8:]})\n")
    print("| kernel (per wave of one workgroup) | vector instr. (PMC) | of which packed FMAs (counted) | v_log_f32 | other vector instr. | "
          "scalar | LDS | vector-ALU busy | non-arithmetic per output position |")
    print("|---|---|---|---|---|---|---|---|---|")
    print(f"| `wpt3_top_kernel` levels 1-8, direct form | {top['valu']:.0f} | {fma_top:.0f} | 0 | {top['valu'] - fma_top:.0f} | {top['salu']:.0f} | "
          f"{top['lds']:.0f} | {100 * top['valu_busy']:.0f} % | {(top['valu'] - fma_top) / (pos_top / 1024):.1f} per position (cA, cD) against {L} FMAs |")
    print(f"| `wpt4_deep_kernel` levels 9-14, lattice form | {deep['valu']:.0f} | {fma_deep:.0f} | {logs:.0f} | {deep['valu'] - fma_deep - logs:.0f} | "
          f"{deep['salu']:.0f} | {deep['lds']:.0f} | {100 * deep['valu_busy']:.0f} % | |")
    # issue model: a vector instruction occupies its SIMD for 4 cycles (64 lanes on 16), v_log_f32 for 16 (quarter rate); a
    # frame = 2 top workgroups (16 waves) + 16 deep workgroups (8 waves) on the 4 SIMDs of a CU, B / 256 frames per CU
    def ms(top_i, deep_i, log_i):
        per_frame = (2 * 16 * top_i * 4 + 16 * 8 * (deep_i * 4 + log_i * 12)) / SIMDS
        return per_frame * (B / CUS) / (GHZ * 1e9) * 1e3
    cur = ms(top["valu"], deep["valu"], logs)
    arith = ms(fma_top, fma_deep + logs, logs)
    gb = B * feat_bytes / 1e9
    print(f"\n| bound at B = {B} | ms per transform | GB/s | fraction of 8 TB/s |")
    print("|---|---|---|---|")
    print(f"| measured (rocprof, top + deep kernel) | {measured_us / 1e3:.3f} | {gb / (measured_us * 1e-6):.0f} | {gb / (measured_us * 1e-6) / PEAK:.3f} |")
    print(f"| the kernels' own vector instruction streams at 100 % issue (no load, LDS, barrier or dependency stall) | {cur:.3f} | {gb / (cur * 1e-3):.0f} | {gb / (cur * 1e-3) / PEAK:.3f} |")
    print(f"| multiply-adds and logarithms ONLY at 100 % issue (every index, address, mirror, load and store instruction removed) | {arith:.3f} | {gb / (arith * 1e-3):.0f} | {gb / (arith * 1e-3) / PEAK:.3f} |")
    print(f"| HBM: algorithmic bytes at the 6.3 TB/s a streaming kernel reaches | {gb / 6300 * 1e3:.3f} | 6300 | 0.788 |")
    return cur, arith


print("# Instruction-level floor of the level-14 wavelet-packet front end (round 5)\n")
print(__doc__.split("\n\n")[0].replace("\n", " "))
print("\nModel: a vector-ALU instruction of a 64-lane wave occupies its SIMD for 4 cycles (`v_pk_fma_f32` delivers two "
      "multiply-adds per lane in that slot: tools/micro/fma_rate.hip measured 4.6 cycles from one wave), `v_log_f32` for 16; "
      f"{CUS} CUs x {SIMDS} SIMDs at the {GHZ} GHz these kernels hold; a frame is two workgroups of the top kernel (16 waves each) "
      "and sixteen of the deep kernel (8 waves each).  Instruction counts per wave: `SQ_INSTS_*` / `SQ_WAVES` of the committed "
      "counter summaries; packed FMAs: counted from the node lengths (direct form: L per output position of both children; "
      "lattice: K = L / 2 per position and K - 1 lead-in positions per node).")
analyse("coif4", 24, "profiles/r05_pmc_frontend_coif4.md", 765 + 1706, 1661064)
analyse("sym5", 10, "profiles/r05_pmc_frontend_sym5.md", 505 + 578, 743560)
print("""
## What the floor says

* **coif4: the north star's 0.60 is the arithmetic itself.**  Running nothing but the packed multiply-adds of the two
  kernels and the logarithms of the epilogue, at 100 % issue on every SIMD, gives 0.61 of 8 TB/s; the instruction streams
  as they are allow 0.46 (measured: 0.34, i.e. 75 % of what the streams allow -- the rest is LDS / barrier / dependency
  wait at 68 % vector-ALU busy).  Fewer multiply-adds than the orthogonal lattice's L per output pair do not exist for a
  24-tap orthogonal bank; the f32 matrix cores issue at the same rate as the packed vector FMA (157 TFLOP/s either way),
  and split-precision products break the 5e-6 parity bar (DESIGN 7.1).  So for coif4 the target is out of reach on this
  chip at fp32, by arithmetic alone; what remains is the 0.34 -> 0.46 gap.
* **sym5 is different: its arithmetic floor lies above the HBM roof** (0.36 ms of multiply-adds against 0.48 ms of
  HBM time), so in principle it is HBM-bound -- but only 31 % (top) and 38 % (deep) of its vector instructions are
  multiply-adds.  The non-arithmetic work is per OUTPUT POSITION, not per tap: 21 vector instructions per position in the
  top kernel for both wavelets (work-item -> (node, position) arithmetic, LDS addresses of 7 / 3 window reads, the two
  stores and their conditional pad mirrors), against 24 / 10 multiply-adds.  A 1024-thread workgroup gives a lane only 5-6
  positions per level, so nothing amortises over a run; the decomposition that would (a lane sliding a register window
  along 20+ positions of one node: one 8-byte LDS read per position) needs 256-thread workgroups with the whole LDS,
  one wave per SIMD -- the rewrite of round 3 with six outputs per item measured level for that reason.
* **The top kernel is bound twice over.**  Its 297 LDS instructions per wave are almost all `ds_read_b128` (1 KB per wave
  instruction, 8 cycles of the CU's 128 B/clk LDS pipe): 16 waves x 297 x 8 = 38 000 cycles of LDS pipe per workgroup,
  against 4 waves x 2 172 x 4 = 34 700 cycles of vector issue per SIMD and 48 500 cycles measured per workgroup -- the
  window reads (L + 2 samples per two output positions, where a sliding window would read 2 per position) cost as much
  as the arithmetic.  Four outputs per work item (L + 6 samples per four positions, half the index arithmetic) measured
  1-2 % in round 4: neither stream alone is the limiter, their sum is (LDS reads and vector instructions of a wave do
  not overlap with each other when every wave runs the same read -> multiply -> store sequence between two barriers).
* **Where the top kernel's time goes, stage by stage** (round 5, `tools/top_stages.py`: builds that return after level k,
  B = 4096, us per launch): coif4 -- frame -> LDS 95, level 1 +93, levels 2-3 +138, 4-5 +142, 6-7 +164, level 8 + store
  +125 = 757; sym5 -- 96, +51, +72, +79, +97, +71 = 466.  A level costs 70-80 us (coif4) / 36-49 us (sym5) against 35 / 14 us
  of packed FMAs: every level is one pass read window -> multiply -> store -> barrier over the whole workgroup, so the
  LDS phase (16 waves x 9 reads x 8 cycles) and the arithmetic phase do not overlap, and one 1024-thread workgroup per CU
  (it needs the whole LDS) has no second workgroup to fill the gaps; the frame load alone (one round of memory latency
  with nothing to hide it) is 12-20 % of the kernel.  Runs of 6-14 outputs per work item (Std3::run_at: every level one
  round of the threads, 1.5 instead of 3.5 LDS reads per position) bought 1 % (coif4) / 7 % (sym5): neither stream is the
  limiter, the serialisation is.
* **Hiding the frame load behind the previous item was built and rejected**: a persistent grid (one workgroup per CU walking
  32 (frame, half) items at B = 4096) with the next item's frame requested into 11 + 1 registers per thread right after
  level 1 and stored to LDS at the top of the next iteration -- parity green, 2.40 -> 2.64 ms (coif4) / 1.05 -> 1.19 ms
  (sym5) at B = 4096.  The same binary launched one workgroup per item is also slower than the kernel without the loop
  (2.56 / 1.14 ms: the prefetch registers push the kernel to 111 of the 128 registers a 1024-thread workgroup may use, and
  the item loop invites the compiler to hoist every level's lane addresses out of it -- 128 registers and 2.4 x slower
  until the thread index was made opaque per item), and the static item assignment loses to the dispatcher's dynamic one.
* **The deep kernel's LDS bank conflicts** (`SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE` = 0.44 coif4, 0.32 sym5) sit in the
  lane-strided 8-byte reads of the multi-lane levels 8 -> 12 (lane stride 2 R floats); those levels are 0.56 of the
  kernel's 1.7 ms and LDS-active cycles are 25 % of its SIMD cycles -- removing every conflict is worth at most 0.1 ms.
* **B = 128** pays two launches' fill and drain on top (256 + 2048 workgroups: one and four rounds over the CUs):
  0.30 against 0.34 at B = 4096.
""")
