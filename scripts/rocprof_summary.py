"""Turn a rocprofv3 results .db (rocpd sqlite) into the per-kernel summary table kept under profiles/."""
import sqlite3
import sys


def main(db_path, out_path, note=""):
    db = sqlite3.connect(db_path)
    rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    total = sum(r[2] for r in rows)
    with open(out_path, "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats summary\n\n%s\n\n" % note)
        f.write("total GPU kernel time: %.3f ms over %d kernels names\n\n" % (total / 1e3, len(rows)))
        f.write("| kernel | calls | total (us) | avg (us) | % |\n|---|---:|---:|---:|---:|\n")
        for name, calls, tot, avg, pct in rows:
            f.write("| `%s` | %d | %.1f | %.2f | %.2f |\n" % (name[:150], calls, tot, avg, pct))
    print("wrote", out_path, "rows", len(rows))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], " ".join(sys.argv[3:]))
