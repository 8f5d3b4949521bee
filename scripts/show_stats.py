"""Prints the top rows of a rocprofv3 kernel_stats.csv (name, calls, average and maximum microseconds, share)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
for r in rows[:n]:
    print("%-70s %5s calls  avg %9.1f us  max %9.1f us  %5.1f %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                                   float(r["MaxNs"]) / 1e3, float(r["Percentage"])))
