"""Which python lines of the c3 training step launch the torch library kernels (fills, adds, copies, small reductions)?

The kernel trace (`profiles/r04_train_full_c3_kernel_trace.txt`) shows ~110 torch/rocm library launches per step beside
the hand-written kernels; this script runs the same step (bench.train_full_section's closure, captured through
bench.timed) under torch.profiler with python stacks and prints, per aten op that launched a kernel, the repo frames it
came from.  Diagnostic only.

    python scripts/step_glue_profile.py [out.txt]
"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def dispatch_trace(step, repo):
    import traceback

    from torch.utils._python_dispatch import TorchDispatchMode

    quiet = ("view", "reshape", "detach", "alias", "expand", "slice", "select", "unsqueeze", "squeeze", "t.default", "permute",
             "transpose", "as_strided", "_unsafe_view", "unbind", "split", "empty", "_local_scalar_dense", "lift_fresh",
             "is_same_size", "stride", "size", "_to_copy", "unfold", "chunk", "narrow")
    seen = collections.Counter()

    class Mode(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            name = str(func)
            if not any(q in name for q in quiet):
                fr = [f for f in traceback.extract_stack() if (repo in f.filename or f.filename.endswith("bench.py"))
                      and "step_glue_profile" not in f.filename]
                where = " <- ".join(f"{f.filename.replace(repo + '/', '')}:{f.lineno}" for f in reversed(fr[-3:]))
                seen[(name, where)] += 1
            return func(*args, **(kwargs or {}))

    with Mode():
        step()
    torch.cuda.synchronize()
    return [f"{c:4d}  {name:34s} {where}" for (name, where), c in sorted(seen.items(), key=lambda kv: (kv[0][1], kv[0][0]))]


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else None
    device = torch.device("cuda:0")
    captured = {}
    orig = bench.timed

    def hook(step, steps, warmup, world, dev):
        captured["step"] = step
        return orig(step, steps, warmup, world, dev)

    bench.timed = hook
    bench.train_full_section(device, 0, 1, steps=3, warmup=3)
    step = captured["step"]
    from torch.profiler import ProfilerActivity, profile

    n = 3
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        for _ in range(n):
            step()
        torch.cuda.synchronize()
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rows = collections.defaultdict(lambda: [0, 0.0])
    for ev in prof.events():
        if ev.device_type != torch.autograd.DeviceType.CPU or not ev.name.startswith("aten::"):
            continue
        t = sum(k.duration for k in ev.kernels)  # kernels this op launched itself
        if not ev.kernels:
            continue
        frames = [f for f in (ev.stack or []) if repo in f or "bench.py" in f]
        where = " <- ".join(f.replace(repo + "/", "") for f in frames[:3]) or "(autograd engine / no repo frame)"
        r = rows[(ev.name, where)]
        r[0] += len(ev.kernels)
        r[1] += t
    lines = [f"torch library launches in the c3 training step, per step (mean of {n}); sorted by device time"]
    tot_n = tot_t = 0
    for (name, where), (cnt, t) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"{cnt / n:6.1f} launches {t / n:8.1f} us  {name:28s} {where}")
        tot_n += cnt
        tot_t += t
    lines.append(f"total {tot_n / n:.1f} launches, {tot_t / n:.1f} us per step")
    lines += ["", "aten ops dispatched from the calling thread during ONE step (forward, sync, optimizer; the autograd engine's",
              "device thread is not seen), by the innermost repo frames; views and metadata ops launch nothing:"]
    lines += dispatch_trace(step, repo)
    text = "\n".join(lines)
    print(text)
    if out_path:
        with open(out_path, "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    main()
