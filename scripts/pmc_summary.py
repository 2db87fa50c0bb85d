"""rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, csv output) -> per-kernel HBM-side traffic table.

    python scripts/pmc_summary.py <dir with pmc_FETCH_SIZE/ and pmc_WRITE_SIZE/> <out.md> <out.json> [resnet|vgg]

When the passes were run with SCDA_GAN_GRAPH=0 SCDA_WINO_LOG=<dir>/wino_log_<counter>.txt (scripts/collect_profiles.sh does), every
dispatch of the three Winograd kernels is joined with the layer it ran (the library logs one line per launch, in launch order =
dispatch order): per-layer read / write bytes against the launch's ALGORITHMIC bytes, and the dominant class's traffic averaged over
exactly the launches bench.py's in-library profiler sees (the detector's forward + data-gradient launches: batch 1).

Units and corrections (MI355X_MICROARCH.md, HBM section): the counters are in KiB... rocprofv3 reports FETCH_SIZE /
WRITE_SIZE in kilobytes; on gfx950 FETCH_SIZE counts 64 B per 128-byte request, i.e. HALF the bytes read -> doubled here.
That factor was re-checked on this workload's own kernels with a known byte count (see the calibration rows).
Infinity-Cache hits are included in these counters, so "traffic" is what leaves the L2, not strictly HBM."""
import collections
import csv
import json
import os
import re
import sys


def load(d, counter):
    out = collections.defaultdict(list)
    path = os.path.join(d, "pmc_%s" % counter, "pmc_counter_collection.csv")
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            out[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return out


def load_wino(d, counter):
    """-> [(log fields, counter value)] for the Winograd dispatches of one pass, in dispatch order; None without a usable log"""
    log = os.path.join(d, "wino_log_%s.txt" % counter)
    path = os.path.join(d, "pmc_%s" % counter, "pmc_counter_collection.csv")
    if not os.path.exists(log):
        return None
    disp = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and "conv_wino" in r["Kernel_Name"] and "pack" not in r["Kernel_Name"]:
            disp[int(r["Dispatch_Id"])] = disp.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
    lines = [l.split() for l in open(log) if l.strip()]
    if len(lines) != len(disp):
        print("wino log: %d lines vs %d dispatches in %s -- not joined" % (len(lines), len(disp), counter))
        return None
    return [(l, disp[k]) for l, k in zip(lines, sorted(disp))]


def wino_layers(d):
    """per (kind, shape): launches, algorithmic bytes, read bytes, write bytes per launch, executed flop"""
    F, W = load_wino(d, "FETCH_SIZE"), load_wino(d, "WRITE_SIZE")
    if not F or not W or len(F) != len(W):
        return None
    agg = collections.OrderedDict()
    for (l, f), (l2, w) in zip(F, W):
        if l != l2:
            print("wino logs of the two passes differ -- not joined")
            return None
        key = (l[0], int(l[3]), int(l[4]), int(l[5]), int(l[6]), int(l[7]))     # kind, batch, C, H, W, M
        a = agg.setdefault(key, [0, float(l[9]), 0.0, 0.0, float(l[10])])
        a[0] += 1; a[2] += 2.0 * 1024 * f; a[3] += 1024 * w
    return agg


def xcd_floor(kind, batch, C, H, W, M, alg_bytes):
    """What the eight NON-COHERENT per-XCD L2s must fetch at least for a forward / data-gradient launch: every XCD needs the transformed
    filters of the m-tiles it serves and the input of the pixel blocks it serves; splitting the XCDs gm x (8 / gm) over m-tile groups x
    pixel-block runs costs (8 / gm) * U + gm * X at the L2s' memory side (Infinity-Cache hits are counted there), U = 16 * M_pad * C * 4
    bytes of transformed filters (16 / 9 of the raw 3x3 weights the algorithmic figure counts), X the input.  The counter can only be
    compared with THIS floor, not with the algorithmic bytes, once U or X exceeds what one 4 MB L2 keeps."""
    if kind == "wgrad":
        return None
    U = 16.0 * ((M + 63) // 64 * 64) * C * 4
    X = 4.0 * batch * C * H * W
    out = alg_bytes - X - 4.0 * 9 * M * C      # (the result, and for a masked data gradient the activation mask it reads)
    # (an operand that fits beside the other in one L2 is fetched once per XCD that uses it, not once per workgroup: the same formula)
    return min((8 / gm) * U + gm * X for gm in (1, 2, 4, 8) if gm <= max(1, (M + 63) // 64) or gm == 1) + out


def short(name):
    name = re.sub(r"\(.*", "", name.replace("void ", "").replace("scda::", ""))
    return name


def main(d, out_md, out_json, which=None):
    F, W = load(d, "FETCH_SIZE"), load(d, "WRITE_SIZE")
    rows = []
    for k, f in F.items():
        w = W.get(k, [0.0])
        rows.append((short(k), len(f), 2.0 * 1024 * sum(f) / len(f), 1024 * sum(w) / len(w)))
    rows.sort(key=lambda r: -(r[2] + r[3]) * r[1])
    dom_name = "conv_wino_kernel<1|2> (forward + data gradient)"
    dom = [r for r in rows if r[0].startswith("conv_wino_kernel<")]
    if which == "resnet":     # the ResNet-50 C4 configuration's dominant class: the 64-row tiles of its 1x1 convolutions (resnet_config.DOMINANT)
        dom_name = "conv_igemm_glds_kernel<64,*,1,1,1,fwd>"
        dom = [r for r in rows if r[0].startswith("conv_igemm_glds_kernel<64, ") and r[0].endswith("1, 1, 1, false>")]
    elif not dom:     # SCDA_WINOGRAD=0: the direct kernel's 128- / 256-row forward instantiations
        dom_name = "conv_igemm_glds_kernel<128|256,*,3,3,1,fwd>"
        dom = [r for r in rows if r[0].startswith(("conv_igemm_glds_kernel<128, ", "conv_igemm_glds_kernel<256, ")) and r[0].endswith("3, 3, 1, false>")]
    n = sum(r[1] for r in dom)
    dom_fetch = sum(r[2] * r[1] for r in dom) / max(n, 1)
    dom_write = sum(r[3] * r[1] for r in dom) / max(n, 1)
    cal = {r[0]: r for r in rows}
    layers = wino_layers(d) if which != "resnet" else None
    like = None
    if layers:
        # the population bench.py's profiler times inside the region: the detector's (batch-1) forward + data-gradient launches
        sel = [(k, a) for k, a in layers.items() if k[0] != "wgrad" and k[1] == 1]
        n_l = sum(a[0] for _, a in sel)
        like = {"launches": n_l, "algorithmic_bytes_per_launch": round(sum(a[0] * a[1] for _, a in sel) / n_l),
                "read_bytes_per_launch": round(sum(a[2] for _, a in sel) / n_l), "write_bytes_per_launch": round(sum(a[3] for _, a in sel) / n_l)}
        like["traffic_bytes_per_launch"] = like["read_bytes_per_launch"] + like["write_bytes_per_launch"]
        like["traffic_over_algorithmic"] = round(like["traffic_bytes_per_launch"] / like["algorithmic_bytes_per_launch"], 3)
    with open(out_md, "w") as f:
        f.write("# rocprofv3 PMC: L2 memory-side traffic per launch\n\n")
        f.write("Two passes of `python bench.py --steps 2 --warmup 1 --no-cpu-baseline" + (" --config resnet50" if which == "resnet" else "") + "`: `rocprofv3 --pmc FETCH_SIZE --kernel-trace` and\n"
                "`rocprofv3 --pmc WRITE_SIZE --kernel-trace` (counters never combined, no other trace domains).  read = 2 x FETCH_SIZE KB\n"
                "(gfx950 counts 64 B per 128-byte request), write = WRITE_SIZE KB.  Infinity-Cache hits are counted, so these are\n"
                "bytes leaving the L2, an upper bound on HBM bytes.\n\n")
        f.write("Calibration of the x2 on kernels of this run whose byte count is known exactly:\n\n")
        for k, expect in (("adam_kernel", "reads 16 B/param, writes 12 B/param -> read/write = 1.333"),
                          ("axpby_kernel", "reads 2 arrays, writes 1 -> read/write = 2.0"),
                          ("act_bwd_kernel", "reads 2 arrays, writes 1 -> read/write = 2.0"),
                          ("dropout_apply_kernel", "reads 4 B + 1 B mask, writes 4 B -> read/write = 1.25")):
            if k in cal and cal[k][3] > 0:
                f.write("* `%s`: %s; measured (2 x FETCH)/WRITE = %.3f\n" % (k, expect, cal[k][2] / cal[k][3]))
        f.write("\nDominant kernel class `%s` (%d launches): read %.1f MB + write %.1f MB = %.1f MB per launch\n\n"
                % (dom_name, n, dom_fetch / 1e6, dom_write / 1e6, (dom_fetch + dom_write) / 1e6))
        if like:
            f.write("**Like for like** (the %d batch-1 forward + data-gradient launches = what `bench.py`'s `roofline` times; the passes ran with\n"
                    "`SCDA_GAN_GRAPH=0 SCDA_WINO_LOG=...` so that every dispatch is joined with its layer): traffic %.1f MB per launch against\n"
                    "%.1f MB algorithmic = **%.2fx**.\n\n" % (like["launches"], like["traffic_bytes_per_launch"] / 1e6,
                                                              like["algorithmic_bytes_per_launch"] / 1e6, like["traffic_over_algorithmic"]))
            f.write("`8-L2 floor` = what eight non-coherent 4 MB L2s must fetch at least: min over gm of (8 / gm) x transformed filters + gm x input\n"
                    "(+ the result once); these counters sit at the L2s' memory side and count Infinity-Cache hits, so from conv3_x on (filters or\n"
                    "input beyond one L2) the floor, not the algorithmic figure, is what the launch order can reach.\n\n")
            f.write("| Winograd launch (kind, batch, C, H, W, M) | launches | algorithmic MB | read MB | write MB | traffic / algorithmic | 8-L2 floor MB | traffic / floor |\n|---|---:|---:|---:|---:|---:|---:|---:|\n")
            fl_sum = tr_sum = 0.0
            for k, a in layers.items():
                fl = xcd_floor(k[0], k[1], k[2], k[3], k[4], k[5], a[1])
                tr = (a[2] + a[3]) / a[0]
                if fl and k[1] == 1:
                    fl_sum += fl * a[0]; tr_sum += tr * a[0]
                f.write("| %s b%d %d x %d x %d -> %d | %d | %.1f | %.1f | %.1f | %.2f | %s | %s |\n"
                        % (k[0], k[1], k[2], k[3], k[4], k[5], a[0], a[1] / 1e6, a[2] / a[0] / 1e6, a[3] / a[0] / 1e6, tr / a[1],
                           "%.1f" % (fl / 1e6) if fl else "-", "%.2f" % (tr / fl) if fl else "-"))
            if fl_sum:
                like["xcd_floor_bytes_per_launch"] = round(fl_sum / like["launches"])
                like["traffic_over_xcd_floor"] = round(tr_sum / fl_sum, 3)
                f.write("\nOver the same %d launches: traffic / 8-L2 floor = **%.2fx**.\n" % (like["launches"], tr_sum / fl_sum))
            f.write("\n")
        f.write("| kernel | launches | read MB/launch | write MB/launch |\n|---|---:|---:|---:|\n")
        for r in rows[:45]:
            f.write("| `%s` | %d | %.2f | %.2f |\n" % (r[0][:110], r[1], r[2] / 1e6, r[3] / 1e6))
    with open(out_json, "w") as f:
        json.dump({"dominant": {"kernel": dom_name, "launches": n,
                                "read_bytes_per_launch": round(dom_fetch), "write_bytes_per_launch": round(dom_write),
                                "traffic_bytes_per_launch": round(dom_fetch + dom_write)},
                   # the detector's forward + data-gradient launches only (bench.py's timed population), joined per dispatch
                   "dominant_detector_launches": like,
                   "wino_layers": [{"kind": k[0], "batch": k[1], "C": k[2], "H": k[3], "W": k[4], "M": k[5], "launches": a[0],
                                    "algorithmic_bytes": round(a[1]), "read_bytes": round(a[2] / a[0]), "write_bytes": round(a[3] / a[0]),
                                    "executed_flop": a[4]} for k, a in layers.items()] if layers else None,
                   "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; read = 2*FETCH_SIZE KB*1024, write = WRITE_SIZE KB*1024"},
                  f, indent=1)
    print("dominant: read %.1f MB write %.1f MB per launch over %d launches" % (dom_fetch / 1e6, dom_write / 1e6, n))


if __name__ == "__main__":
    main(*sys.argv[1:5])
