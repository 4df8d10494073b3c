"""Summary of a rocprofv3 --kernel-trace CSV: per kernel name the launches, mean duration, and how many launches of that kernel
(and of any kernel) were running at the same time on average -- does the work of concurrent host threads overlap on the device?

    python tools/trace_concurrency.py <dir with *_kernel_trace.csv>
"""
import collections, csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].split("<")[0][:28],
                     r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
by = collections.defaultdict(list)
for a, b, n, q, s in rows:
    by[n].append((a, b))
print(f"{len(rows)} launches over {(t1 - t0) / 1e6:.1f} ms; queues {len(set(r[3] for r in rows))}, streams {len(set(r[4] for r in rows))}")
busy_any = sum(b - a for a, b, *_ in rows)
# union length of all intervals (time with at least one kernel running)
def union(iv):
    tot, cur_a, cur_b = 0, None, None
    for a, b in sorted(iv):
        if cur_b is None or a > cur_b:
            if cur_b is not None: tot += cur_b - cur_a
            cur_a, cur_b = a, b
        else:
            cur_b = max(cur_b, b)
    return tot + (cur_b - cur_a if cur_b is not None else 0)
u = union([(a, b) for a, b, *_ in rows])
print(f"device busy (any kernel) {u / 1e6:.1f} ms = {100.0 * u / (t1 - t0):.0f} % of the span; mean kernels in flight while busy {busy_any / u:.2f}")
print(f"{'kernel':30s} {'launches':>8s} {'mean us':>9s} {'sum ms':>8s} {'own overlap':>11s}")
for n, iv in sorted(by.items(), key=lambda kv: -sum(b - a for a, b in kv[1])):
    s = sum(b - a for a, b in iv)
    print(f"{n:30s} {len(iv):8d} {s / len(iv) / 1e3:9.1f} {s / 1e6:8.1f} {s / max(union(iv), 1):11.2f}")
