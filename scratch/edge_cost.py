"""Cost of an event edge on the recording stream (dependent-kernel chain with / without records in between)."""
import time, torch
dev = torch.device("cuda:0")
x = torch.zeros(16 << 20, device=dev)
side = torch.cuda.Stream()
def chain(n, mode):
    cur = torch.cuda.current_stream()
    for _ in range(n):
        x.add_(1.0)
        if mode == "record":
            ev = torch.cuda.Event(); ev.record(cur)
        elif mode == "edge":
            ev = torch.cuda.Event(); ev.record(cur); side.wait_event(ev)
        elif mode == "edge+work":
            ev = torch.cuda.Event(); ev.record(cur); side.wait_event(ev)
            with torch.cuda.stream(side): y.add_(1.0)
y = torch.zeros(1024, device=dev)
for mode in ("none", "record", "edge", "edge+work", "none"):
    chain(200, mode); torch.cuda.synchronize()
    s0 = torch.cuda.Event(enable_timing=True); s1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); s0.record(); chain(2000, mode); s1.record(); torch.cuda.synchronize()
    print(f"{mode:10s} gpu {s0.elapsed_time(s1)/2000*1e3:7.2f} us/iter   host {(time.perf_counter()-t0)/2000*1e6:7.2f} us/iter", flush=True)
