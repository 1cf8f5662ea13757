import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "aej::" in r["Name"]]
for r in sorted(rows, key=lambda r: -int(r["TotalDurationNs"])):
    print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"])/1e3:10.1f} total_ms {int(r["TotalDurationNs"])/1e6:8.2f}')
