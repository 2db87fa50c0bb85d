"""rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE (csv) -> MFMA pipe utilisation per kernel.
util = SQ_VALU_MFMA_BUSY_CYCLES (summed over the 1024 SIMDs) / (GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 * 1024)"""
import collections, csv, re, sys
d = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    d[r['Kernel_Name']][r['Counter_Name']].append((float(r['Counter_Value']), int(r['End_Timestamp']) - int(r['Start_Timestamp'])))
rows = []
for k, c in d.items():
    if 'SQ_VALU_MFMA_BUSY_CYCLES' not in c or 'GRBM_GUI_ACTIVE' not in c:
        continue
    m = sum(v for v, _ in c['SQ_VALU_MFMA_BUSY_CYCLES']); g = sum(v for v, _ in c['GRBM_GUI_ACTIVE'])
    n = len(c['GRBM_GUI_ACTIVE']); t = sum(t for _, t in c['GRBM_GUI_ACTIVE'])
    if m > 0:
        rows.append((m, re.sub(r"\(.*", "", k.replace("void ", "").replace("scda::", "")), n, m / (g / 8 * 1024), t / n / 1e3, g / 8 / (t / 1e3)))
rows.sort(reverse=True)
with open(sys.argv[2], "w") as f:
    f.write("# rocprofv3 PMC: MFMA pipe utilisation per kernel\n\n"
            "`rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline`\n"
            "(counter pass only, kernels run one at a time).  util = MFMA-busy cycles summed over the 1024 SIMDs / (active cycles per XCD x 1024);\n"
            "for the fp32 `v_mfma_f32_32x32x2_f32` (64 cycles, 4096 FLOP) util x 157.3 TFLOP/s is the achieved rate; for `gemm_x9_kernel` (nine\n"
            "`v_mfma_f32_32x32x16_bf16` of 32 cycles per 32 x 32 x 16 fp32 products) util x 157.3 x 16 / 9 = the fp32-EQUIVALENT rate.  Clock = active cycles /\n"
            "duration (meaningful for launches of >= 100 us; isolated launches of a counter pass run at higher clocks than the iteration holds).\n\n"
            "| kernel | launches | MFMA util | = TFLOP/s | avg us | clock MHz |\n|---|---:|---:|---:|---:|---:|\n")
    for m, k, n, u, us, mhz in rows[:24]:
        f.write("| `%s` | %d | %.1f %% | %.0f | %.1f | %.0f |\n" % (k[:90], n, 100 * u, u * 157.3 * (16.0 / 9.0 if "gemm_x9_kernel" in k else 1.0), us, mhz))
print(open(sys.argv[2]).read()[:2500])
