#!/usr/bin/env python
"""What the engine pool's streams do to each other: phase / concurrency accounting of a pooled rocprofv3 kernel trace.

On the GPU box (the program directly after `--`):
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace -- \
        python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --streams 3 --lite
    python3 tools/pool_timeline.py pack gpurun_out/trace gpurun_out/r5_trace.pkl.gz      # 6 MB csv -> 0.3 MB
Anywhere:
    python3 tools/pool_timeline.py report gpurun_out/r5_trace.pkl.gz > profiles/r05_pool_timeline.txt

The report takes the longest busy segment of the trace (the timed steps) and prints: how long only image-side kernels, only
decode kernels, or both were running; how many kernels were in flight; per kernel class the mean duration next to the gap in
front of it inside its own queue; every queue's encode / decode-step segments; and an excerpt of one decoder layer.
A kernel's Start_Timestamp is its DISPATCH (inside a queue the next kernel starts the instant the previous one ends), so a
duration includes the wait for CUs that kernels of other queues hold."""
import collections
import csv
import glob
import gzip
import os
import pickle
import statistics as st
import sys


def cls(n):
    if "gemm_pp" in n:
        return "enc_gemm"
    if "vit_attention" in n:
        return "vit_attn"
    if "layernorm_kernel" in n and "reduce" not in n:
        return "enc_ln"
    if "gemm_rows" in n or "dec_tile" in n or "dec_small_gemm" in n:
        return "dec_gemm"
    if "decode_attention_online" in n or "decode_attention_shared" in n or "dec_cross" in n or "dec_small_cross" in n:
        return "cross"
    if "reduce_layernorm" in n:
        return "red_ln"
    if "decode_attention_wave" in n:
        return "self"
    return "other"


ENC = ("enc_gemm", "vit_attn", "enc_ln")
DEC = ("dec_gemm", "cross", "red_ln", "self")


def pack(src, dst):
    f = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = [(r["Kernel_Name"][:90], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id"))
            for r in csv.DictReader(open(f))]
    pickle.dump(rows, gzip.open(dst, "wb"))
    print(f"{len(rows)} dispatches -> {dst}")


def report(path):
    rows = pickle.load(gzip.open(path, "rb"))
    rows = [tuple(r[:4]) for r in rows]
    rows.sort(key=lambda r: r[1])
    segs, s, e = [], rows[0][1], rows[0][2]
    for r in rows[1:]:
        if r[1] > e + 2e6:
            segs.append((s, e)); s = r[1]
        e = max(e, r[2])
    segs.append((s, e))
    A, B = max(segs, key=lambda x: x[1] - x[0])
    R = [r for r in rows if r[1] >= A and r[2] <= B]
    nq = len({r[3] for r in R})
    print(f"timed segment: {(B - A) / 1e6:.2f} ms, {len(R)} dispatches on {nq} queues")
    ev = []
    for r in R:
        c = cls(r[0]); ev.append((r[1], 1, c)); ev.append((r[2], -1, c))
    ev.sort()
    active, last = collections.Counter(), None
    phase, nact, combo = collections.Counter(), collections.Counter(), collections.Counter()
    for t, d, c in ev:
        if last is not None:
            n = sum(active.values())
            enc, dec = sum(active[k] for k in ENC), sum(active[k] for k in DEC)
            phase["idle" if n == 0 else "image side only" if dec == 0 else "decode only" if enc == 0 else "both"] += t - last
            nact[n] += t - last
            combo[tuple(sorted((k, v) for k, v in active.items() if v > 0))] += t - last
        active[c] += d; last = t
    tot = B - A
    for k, v in sorted(phase.items(), key=lambda kv: -kv[1]):
        print(f"  {k:16s} {v / 1e6:8.2f} ms  {100 * v / tot:5.1f} %")
    print("kernels in flight: " + ", ".join(f"{k}: {v / 1e6:.1f} ms" for k, v in sorted(nact.items())))
    print("most frequent combinations:")
    for k, v in sorted(combo.items(), key=lambda kv: -kv[1])[:10]:
        print(f"  {100 * v / tot:5.1f} %  {v / 1e6:7.2f} ms  {k}")
    byq = collections.defaultdict(list)
    for r in R:
        byq[r[3]].append(r)
    gap, dur = collections.defaultdict(list), collections.defaultdict(list)
    for rs in byq.values():
        for a, b in zip(rs, rs[1:]):
            gap[cls(b[0])].append((b[1] - a[2]) / 1e3); dur[cls(b[0])].append((b[2] - b[1]) / 1e3)
    print("per class (inside its own queue): launches, mean / median duration us, mean gap in front us, total ms")
    for k in sorted(dur, key=lambda k: -sum(dur[k])):
        print(f"  {k:9s} {len(dur[k]):6d}  {st.mean(dur[k]):8.2f} {st.median(dur[k]):8.2f}  {st.mean(gap[k]):6.2f}  {sum(dur[k]) / 1e3:8.1f}")
    for q, rs in sorted(byq.items()):
        out, cur = [], None
        for r in rs:
            c = cls(r[0])
            typ = "D" if c in DEC else "E" if c in ENC else None
            if typ is None:
                continue
            if cur and cur[0] == typ:
                cur[2] = r[2]; cur[3] += 1
            else:
                if cur:
                    out.append(cur)
                cur = [typ, r[1], r[2], 1]
        if cur:
            out.append(cur)
        big = [s for s in out if s[3] > 30]
        if not big:
            continue
        enc = [s for s in big if s[0] == "E"]
        dec = [s for s in big if s[0] == "D"]
        print(f"queue {q}: " + "; ".join(f"encode {(s[1] - A) / 1e6:.1f}-{(s[2] - A) / 1e6:.1f} ms" for s in enc))
        if dec:
            print(f"          {len(dec)} decode steps, mean {st.mean((s[2] - s[1]) / 1e6 for s in dec):.2f} ms each ({dec[0][3]} kernels)")
    # one decoder layer of the middle of the segment, all queues
    mid = A + (B - A) * 0.36
    ex = [r for r in R if r[1] >= mid and cls(r[0]) in DEC][:66]
    print("excerpt (us from its first dispatch): queue, class, start, end, duration")
    for r in ex:
        print(f"  q{r[3]} {cls(r[0]):8s} {(r[1] - ex[0][1]) / 1e3:8.2f} {(r[2] - ex[0][1]) / 1e3:8.2f} {(r[2] - r[1]) / 1e3:7.2f}")


if __name__ == "__main__":
    if len(sys.argv) >= 4 and sys.argv[1] == "pack":
        pack(sys.argv[2], sys.argv[3])
    elif len(sys.argv) >= 3 and sys.argv[1] == "report":
        report(sys.argv[2])
    else:
        raise SystemExit(__doc__)
