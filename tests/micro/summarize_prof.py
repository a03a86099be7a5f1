"""Post-process rocprofv3 CSV output on the GPU box into the small files that get committed under profiles/.

  python tests/micro/summarize_prof.py stats <dir> <out.csv>            # copy the kernel-stats table (top rows)
  python tests/micro/summarize_prof.py pmc <out.json> <kernel substring> <dir> [<dir> ...]
      mean counter value per launch of the matching kernel, one entry per counter found in the directories
"""
import csv
import glob
import json
import os
import sys


def find(d, pattern):
    return sorted(glob.glob(os.path.join(d, "**", pattern), recursive=True))


def stats(d, out):
    files = find(d, "*kernel_stats.csv")
    if not files:
        raise SystemExit(f"no kernel_stats.csv under {d}")
    rows = list(csv.reader(open(files[0])))
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        for r in rows[:12]:
            r = list(r)
            if r and len(r[0]) > 160:
                r[0] = r[0][:157] + "..."
            w.writerow(r)


def pmc(out, kernel, dirs):
    res = {"kernel_substring": kernel, "counters": {}}
    for d in dirs:
        for path in find(d, "*counter_collection.csv"):
            rd = csv.DictReader(open(path))
            acc = {}
            for row in rd:
                if kernel not in row.get("Kernel_Name", ""):
                    continue
                name, val = row["Counter_Name"], float(row["Counter_Value"])
                key = (row.get("Dispatch_Id"), name)
                acc[key] = acc.get(key, 0.0) + val                 # (one row per XCD / dimension: sum them per dispatch)
            per = {}
            for (_, name), v in acc.items():
                per.setdefault(name, []).append(v)
            for name, vals in per.items():
                res["counters"][name] = {"launches": len(vals), "mean_per_launch": sum(vals) / len(vals),
                                         "min": min(vals), "max": max(vals), "source": os.path.relpath(path)}
    with open(out, "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4:])
