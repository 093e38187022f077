"""Filter a rocprofv3 --kernel-trace --stats CSV down to this repo's kernels (k_*) -> markdown table."""
import csv
import sys


def main(path, title):
    rows = [r for r in csv.DictReader(open(path)) if r["Name"].lstrip("void ").startswith("k_")]
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("### %s\n" % title)
    print("| kernel | calls | avg us | min us | max us | share of k_* time |")
    print("|---|---|---|---|---|---|")
    for r in rows:
        name = r["Name"].replace("void ", "").split("(")[0]
        print("| %s | %s | %.1f | %.1f | %.1f | %.1f %% |" % (
            name, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3,
            100 * float(r["TotalDurationNs"]) / tot))
    print()


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else sys.argv[1])
