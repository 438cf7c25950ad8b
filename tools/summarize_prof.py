"""Shortens a rocprofv3 --kernel-trace --stats kernel_stats.csv (template names) into a
readable summary kept under profiles/."""
import csv
import re
import sys


def short(name):
    name = re.sub(r"rocprim::ROCPRIM_\d+_NS::detail::", "rocprim::", name)
    m = re.search(r"(radix_sort_onesweep_\w+|k_\w+(<[^>]*>)?|__amd_rocclr_\w+|scan\w*|lookback_scan\w*|init_\w+)", name)
    if "trampoline_kernel" in name and m:
        return "rocprim::" + m.group(1)
    name = re.sub(r"\(.*", "", name)
    return name.replace("void ", "").replace(", ", " ").replace(",", " ")[:80]   # (no commas inside a csv field: template arguments)


def main(src, dst, title):
    rows = list(csv.DictReader(open(src)))
    with open(dst, "w") as f:
        f.write(f"# {title}\n# source: rocprofv3 --kernel-trace --stats --output-format csv (kernel_stats.csv), names shortened\n")
        f.write("kernel,calls,total_ms,avg_us,pct,min_us,max_us\n")
        for r in rows:
            f.write("%s,%s,%.3f,%.1f,%s,%.1f,%.1f\n" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                      float(r["AverageNs"]) / 1e3, r["Percentage"], float(r["MinNs"]) / 1e3,
                                                      float(r["MaxNs"]) / 1e3))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "kernel stats")
