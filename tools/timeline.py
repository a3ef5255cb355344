"""Kernel timeline of the last timed steps from a rocprofv3 --kernel-trace CSV: start and end of every launch relative to the
first one listed (us), stream by stream.  usage: python tools/timeline.py <kernel_trace.csv> [launches]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("asora::", "").split("<")[0]
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:9.1f} {(int(r["End_Timestamp"]) - t0) / 1e3:9.1f}  q{r.get("Queue_Id", "?"):>3}  {name}')
