"""compact view of a bench.py JSON line: python scripts/show_bench.py <file.json>"""
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable:", e); continue
    print(f, {k: d.get(k) for k in ("metric", "value", "ms_per_step", "n_gpus")})
    r = d.get("roofline") or {}
    print("  roofline", {k: r.get(k) for k in ("kernel_ms", "achieved", "frac", "traffic")})
    for k in ("train", "train_full"):
        if d.get(k): print(" ", k, {x: d[k].get(x) for x in ("iters_per_sec", "ms_per_iter", "rays_per_sec", "grad_exchange_bytes_per_rank")})
    for k in ("cpu_baseline", "reference_torch_cpu"):
        if d.get(k): print(" ", k, {x: d[k].get(x) for x in ("value", "cores", "kind")})
    for k in ("parity_rel_l2_vs_oracle", "render_kernel_ms", "rays_per_sec", "proposal_evals_per_sec"):
        if k in d: print(" ", k, d[k] if not isinstance(d[k], dict) else {x: d[k][x] for x in list(d[k])[:2]})
