"""Per-kernel table of derived SQ-counter figures from the two `rocprofv3 --pmc` passes of tools/collect_profiles.sh
(`r5_sq_pmc_<tag>_pass_{a,b}.txt`, written by tools/pmc_by_kernel.py: counter values per launch).
  python tools/sq_table.py profiles/r5_sq_pmc_f32_c1_pass_a.txt profiles/r5_sq_pmc_f32_c1_pass_b.txt [min share of wave cycles, %]
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (x 4), SQ_VALU_MFMA_BUSY_CYCLES cycles (MI355X_MICROARCH.md).
`pipe busy` = MFMA-busy cycles / (4 SIMDs x the CUs the launch occupies) / a wave's lifetime -- meaningful for the persistent /
one-round kernels whose waves live as long as the launch (the GEMMs, dwproj, fused_mid)."""
import re
import sys


def load(path):
    out = {}
    for line in open(path):
        m = re.match(r"(.{64}) calls\s+(\d+)\s+(.*)$", line.rstrip("\n"))
        if not m:
            continue
        vals = m.group(3).split()
        out[m.group(1).strip()] = (int(m.group(2)), {vals[i]: float(vals[i + 1]) for i in range(0, len(vals), 2)})
    return out


a, b = load(sys.argv[1]), load(sys.argv[2])
min_share = float(sys.argv[3]) if len(sys.argv) > 3 else 0.4
tot = sum(c * v["SQ_WAVE_CYCLES"] for c, v in a.values() if "SQ_WAVE_CYCLES" in v and not c == 0)
rows = []
for k, (calls, va) in a.items():
    if k not in b or k.startswith(("__amd", "at::")) or "SQ_WAVE_CYCLES" not in va:
        continue
    vb = b[k][1]
    wc = va["SQ_WAVE_CYCLES"]
    if wc <= 0 or calls * wc / tot * 100 < min_share:
        continue
    waves = vb["SQ_WAVES"]
    cyc = 4 * wc / waves
    wgs = waves / (8 if "dwproj_kernel<0, 2, 4" in k or "dwproj_kernel<3, 2, 4" in k or "dwproj_kernel<0, 4, 2" in k or "dwproj_kernel<3, 4, 2" in k
                   or "h16_dma_kernel<2, 4" in k else 4)
    cus = min(256.0, wgs)
    rows.append((calls * wc, k, calls, waves, cyc, va["SQ_WAIT_ANY"] / wc, va["SQ_WAIT_INST_ANY"] / wc, va["SQ_ACTIVE_INST_ANY"] / wc,
                 va["SQ_WAIT_INST_LDS"] / wc, va["SQ_INSTS_VALU"] / vb["SQ_INSTS_MFMA"] if vb["SQ_INSTS_MFMA"] else float("nan"),
                 vb["SQ_LDS_BANK_CONFLICT"] / vb["SQ_LDS_IDX_ACTIVE"] if vb["SQ_LDS_IDX_ACTIVE"] else 0.0,
                 va["SQ_VALU_MFMA_BUSY_CYCLES"], va["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * cus) / cyc if wgs <= 512 else float("nan"),
                 vb["SQ_INSTS_VMEM"] / waves))
print("| kernel | launches | waves per launch | cycles per wave | in s_waitcnt / barrier | in issue stalls | issuing | LDS issue stall | "
      "VALU per MFMA | LDS conflict / active | MFMA-busy cycles per launch | pipe busy | VMEM instr per wave |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
for r in sorted(rows, reverse=True):
    _, k, calls, waves, cyc, w, st, act, lds, vpm, conf, busy, pb, vm = r
    print("| `%s` | %d | %.0f | %.0f | %.0f %% | %.0f %% | %.0f %% | %.1f %% | %s | %.0f %% | %.3g | %s | %.0f |" % (
        k, calls, waves, cyc, 100 * w, 100 * st, 100 * act, 100 * lds, "-" if vpm != vpm else "%.1f" % vpm, 100 * conf, busy,
        "- (several rounds)" if pb != pb else "%.0f %%" % (100 * pb), vm))
