#!/bin/bash
# MFMA utilisation of the attention kernel: kernel-trace pass + one PMC pass, condensed into gpurun_out/attn_profile.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/attn_prof; rm -rf $OUT; mkdir -p $OUT
python $R/tools/attn_bench.py > $OUT/plain.txt 2>&1
for i in 0 1 2 3 4 5 6 7 8; do
  REPS=5 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt$i -- python $R/tools/attn_bench.py $i > $OUT/kt$i.log 2>&1
  REPS=5 timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/pmc$i -- python $R/tools/attn_bench.py $i > $OUT/pmc$i.log 2>&1
  REPS=5 timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VALU SQ_INSTS_LDS --output-format csv -d $OUT/pmcb$i -- python $R/tools/attn_bench.py $i > $OUT/pmcb$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, os, re
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); out = R + "/gpurun_out/attn_prof"
plain = [l for l in open(out + "/plain.txt") if l.startswith("CASE")]
lines = ["MFMA attention kernel (td::attn_mfma_kernel) on MI355X; peak 2500 TFLOP/s dense bf16.  useful = 4*B*H*Lq*Lk*D FLOP; issued = MFMA FLOP incl. the padding of d to 16 (QK^T) / 32 (PV) and of the tiles.",
         "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs) over the kernel's dispatches (guide: busy cycles = 32 per 32x32x16 MFMA).", ""]
for i, pl in enumerate(plain):
    dur = []
    for f in glob.glob(f"{out}/kt{i}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "attn_mfma_kernel" in r["Kernel_Name"]: dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    cnt = {}
    for f in glob.glob(f"{out}/pmc{i}/**/*counter_collection.csv", recursive=True) + glob.glob(f"{out}/pmcb{i}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "attn_mfma_kernel" in r["Kernel_Name"]:
                a = cnt.setdefault(r["Counter_Name"], [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
    m = re.search(r"useful GFLOP ([\d.]+) \| issued MFMA GFLOP ([\d.]+)", pl)
    useful, issued = float(m.group(1)), float(m.group(2))
    if dur:
        dur = sorted(dur)[1:] or dur
        us = sum(dur) / len(dur)
        g = lambda c: cnt[c][1] / cnt[c][0] if c in cnt else None
        busy = g("SQ_VALU_MFMA_BUSY_CYCLES") / (g("GRBM_GUI_ACTIVE") / 8 * 1024) if g("SQ_VALU_MFMA_BUSY_CYCLES") and g("GRBM_GUI_ACTIVE") else float("nan")
        wait = g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES") if g("SQ_WAIT_ANY") and g("SQ_WAVE_CYCLES") else float("nan")
        vpm = ""
        if g("SQ_INSTS_VALU") and g("SQ_INSTS_VALU_MFMA_MOPS_BF16"):
            mf = g("SQ_INSTS_VALU_MFMA_MOPS_BF16") / 64.0   # MOPS counter: 64 per 32x32x16 bf16 MFMA
            vpm = f" | {(g('SQ_INSTS_VALU') - mf) / mf:.1f} non-MFMA VALU per MFMA"
            if g("SQ_ACTIVE_INST_VALU") and g("SQ_WAVE_CYCLES"): vpm += f", VALU issue active {100 * g('SQ_ACTIVE_INST_VALU') / g('SQ_WAVE_CYCLES'):.0f} % of wave cycles"
        lines.append(pl.strip().split(" | wall")[0] + vpm + f" | kernel {us:.1f} us | useful {useful / us * 1e3:.1f} TFLOP/s = {useful / us * 1e3 / 25:.1f} % of peak | issued {issued / us * 1e3:.1f} TFLOP/s = {issued / us * 1e3 / 25:.1f} % | mfma_busy {100 * busy:.1f} % | waves parked {100 * wait:.0f} %")
    else:
        lines.append(pl.strip() + " | (no kernel rows found)")
open(R + "/gpurun_out/attn_profile.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
rm -rf $OUT/kt* $OUT/pmc*
