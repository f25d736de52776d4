"""Which host statements wait for the stream?  (GPU box) — each probe runs behind ~50 ms of queued GEMMs."""
import time, torch
x = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
scal = torch.zeros(6, device="cuda")
loss = torch.zeros((), device="cuda", dtype=torch.float16)
cnt = torch.zeros(4, device="cuda", dtype=torch.int32)
def busy():
    y = x
    for _ in range(60):
        y = y @ x
    return y
def probe(name, fn):
    fn(); torch.cuda.synchronize()        # first use loads the kernel's code object, which waits for the device
    torch.cuda.synchronize(); busy(); t = time.perf_counter(); fn(); dt = time.perf_counter() - t; torch.cuda.synchronize()
    print(f"{name:40s} host {dt*1e3:8.2f} ms")
probe("nothing", lambda: None)
probe("scal[0] = loss.float()", lambda: scal.__setitem__(0, loss.detach().float()))
probe("scal[1] = 40.0", lambda: scal.__setitem__(1, 40.0))
probe("torch.stack 0-dim views", lambda: torch.stack([cnt[0], cnt[1], cnt[2], cnt[3]]).float())
probe("scal[2:6] = gpu tensor", lambda: scal.__setitem__(slice(2, 6), cnt.float()))
probe("acc[1] += 40.0", lambda: scal[1].add_(40.0))
probe("scal[1].fill_(40.0)", lambda: scal[1].fill_(40.0))
