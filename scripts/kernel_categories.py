"""kernel_stats.md (scripts/rocprof_summary.py) -> per-category launches / time per iteration"""
import re, sys
path, iters = sys.argv[1], float(sys.argv[2])
rows = []
for l in open(path):
    m = re.match(r"\| `(.*)` \| (\d+) \| ([\d.]+) \| ([\d.]+) \| ([\d.]+) \|", l)
    if m:
        rows.append((m.group(1), int(m.group(2)), float(m.group(3))))
def cat(k):
    if 'conv_wino_wgrad' in k: return 'conv wgrad (Winograd MFMA)'
    if 'conv_wino_kernel' in k: return 'conv fwd/dgrad (Winograd MFMA)'
    if 'conv_igemm_glds' in k: return 'conv fwd/dgrad (direct-to-LDS MFMA)'
    if 'conv_igemm_kernel' in k: return 'conv gather (MFMA)'
    if 'conv_wgrad' in k: return 'conv wgrad (MFMA)'
    if 'gemm_x9_fixup' in k: return 'split-K reduce'
    if 'gemm_x9' in k: return 'dense GEMM (bf16 x 9 MFMA)'
    if 'gemm_' in k: return 'dense GEMM (MFMA)'
    if 'splitk_reduce' in k: return 'split-K reduce'
    if 'at::native' in k or 'rocclr' in k: return 'torch / runtime (fill, copy, add, cat ...)'
    return re.sub(r"\(.*", "", k.replace('scda::', '').replace('void ', ''))[:40]
agg = {}
for k, n, t in rows:
    a = agg.setdefault(cat(k), [0, 0.0]); a[0] += n; a[1] += t
tot = sum(v[1] for v in agg.values())
print("| category | launches / iteration | ms / iteration | % |\n|---|---:|---:|---:|")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("| %s | %.1f | %.3f | %.1f |" % (k, v[0] / iters, v[1] / iters / 1000, 100 * v[1] / tot))
print("| **total** | **%.0f** | **%.2f** | |" % (sum(v[0] for v in agg.values()) / iters, tot / iters / 1000))
