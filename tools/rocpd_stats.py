"""Summarise kernel durations from a rocprofv3 rocpd (.db) or kernel-trace CSV into a text table.

    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [> profiles/r01_x.txt]
    python tools/rocpd_stats.py x_results.db busy [marker kernel, one launch per step; default optim_update_kernel]
    python tools/rocpd_stats.py x_results.db categories
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = cur.execute(
        f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
        f"from kernels group by {name_col} order by sum(end-start) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print(f"{'kernel':<90} {'calls':>7} {'total_ms':>10} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'pct':>6}")
    for n, c, s, a, mn, mx in rows:
        print(f"{n[:90]:<90} {c:>7} {s/1e6:>10.3f} {a/1e3:>10.2f} {mn/1e3:>10.2f} {mx/1e3:>10.2f} {100*s/tot:>6.2f}")


if __name__ == "__main__" and len(sys.argv) == 2:
    main(sys.argv[1])


def categories(path):
    """Coarse time split of a trace: our HIP kernels / hipBLASLt GEMMs / attention / other torch kernels."""
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    cat = {}
    for n, d in cur.execute(f"select {name_col}, end-start from kernels"):
        if "vsde::attn_" in n:
            k = "vsde attention"
        elif "vsde::" in n:
            k = "vsde " + ("head/gemm/elbo" if any(x in n for x in ("head_", "::tn_", "gemm_nt", "elbo", "pack_")) else "encoder fused")
        elif "Cijk_" in n:
            k = "hipBLASLt GEMM"
        elif n in ("attn_fwd", "bwd_kernel_dk_dv", "bwd_kernel_dq", "bwd_preprocess", "bwd_postprocess") or "fmha_" in n:
            k = "attention (aotriton / aiter)"
        else:
            k = "other torch kernels"
        cat[k] = cat.get(k, 0) + d
    tot = sum(cat.values())
    for k, v in sorted(cat.items(), key=lambda kv: -kv[1]):
        print(f"{k:<32} {v/1e6:>10.3f} ms {100*v/tot:>6.2f}%")


def busy(path, marker="optim_update_kernel", first=35, last=65):
    """GPU busy fraction between the `first`-th and `last`-th launch of the marker kernel (one per training step; the defaults sit
    inside the 50 timed steps of a default bench.py run: 3 + 4 probe steps, 3 capture warm-ups, 4 replays, 10 warm-up steps come first):
    sum of kernel durations / wall span, plus the per-step wall time and the number of launches per step."""
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = cur.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
    marks = [i for i, r in enumerate(rows) if marker in r[0]]
    if len(marks) <= last:
        # short runs (the synthetic workload is profiled with --steps 5 --warmup 2): the last timed steps of whatever there is --
        # the final 4 markers belong to the per-kernel timing steps and the profiled eager step that follow the timed region
        if len(marks) < 8:
            print("not enough marker launches:", len(marks)); return
        last = len(marks) - 5
        first = max(last - 4, len(marks) // 2)
        if last <= first:
            print("not enough marker launches:", len(marks)); return
    a, b = marks[first], marks[last]
    t0, t1 = rows[a][1], rows[b][1]
    dur = sum(r[2] - r[1] for r in rows[a:b])
    # union of busy intervals (kernels may overlap)
    busy_ns, cur_end = 0, t0
    for _, s, e in rows[a:b]:
        if e > cur_end:
            busy_ns += e - max(s, cur_end); cur_end = e
    steps = last - first
    print(f"steps {steps}: wall {1e-6 * (t1 - t0) / steps:.3f} ms/step, sum of kernel durations {1e-6 * dur / steps:.3f} ms/step, "
          f"GPU busy (union) {100.0 * busy_ns / (t1 - t0):.1f} %, launches/step {(b - a) / steps:.0f}")
    gaps = sorted(((rows[i + 1][1] - rows[i][2]) for i in range(a, b - 1)), reverse=True)
    print("idle gaps: total %.3f ms/step; gaps > 5 us: %d per step; largest (us): %s" % (
        1e-6 * sum(g for g in gaps if g > 0) / steps, sum(1 for g in gaps if g > 5000) / steps, [round(g / 1e3, 1) for g in gaps[:8]]))


if __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[2] == "busy":
    busy(sys.argv[1], *(sys.argv[3:4]))
elif __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[2] == "categories":
    categories(sys.argv[1])
