"""The kernel trace of a bench run (rocprofv3 --kernel-trace --output-format csv) laid out by queue: for the last steps, every kernel with its start (ms from the
window's begin), duration and the idle gap in front of it on its queue; then busy time and gaps per queue.
   python tools/trace_timeline.py <dir of the trace> [window in ms, default 700] [min duration in ms to list, default 0.5]"""
import csv, glob, sys, re, collections
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
win = float(sys.argv[2]) if len(sys.argv) > 2 else 700.0
mind = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id', r.get('Stream_Id', '?')), r['Kernel_Name']))
rows.sort()
# the window: the last `win` ms before the last k_dp / k_pair kernel of the run's main loop (the extras that follow a bench's loops are cut off by the caller's flags)
tend = max(e for s, e, q, n in rows if 'k_dp' in n or 'k_pair' in n or 'k_stitch' in n)
t0 = tend - int(win * 1e6)
def short(n):
    n = re.sub(r'hlala::', '', n); n = re.sub(r'\(.*', '', n); n = re.sub(r'^void ', '', n)
    return n[:44]
byq = collections.defaultdict(list)
for s, e, q, n in rows:
    if e >= t0 and s <= tend: byq[q].append((s, e, n))
for q, v in sorted(byq.items(), key=lambda kv: -sum(e - s for s, e, n in kv[1])):
    busy = sum(min(e, tend) - max(s, t0) for s, e, n in v) / 1e6
    print("== queue %s: %d kernels, busy %.1f ms of %.1f" % (q, len(v), busy, win))
    last = None
    for s, e, n in v:
        d = (e - s) / 1e6
        gap = (s - last) / 1e6 if last is not None else 0.0
        if d >= mind or gap >= 1.0:
            print("   %8.1f  %7.2f ms  %s%s" % ((s - t0) / 1e6, d, short(n), ("   <- %.1f ms idle before" % gap) if gap >= 1.0 else ""))
        last = max(last, e) if last is not None else e
