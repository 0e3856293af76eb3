import time, torch
x = torch.zeros(1024, device="cuda")
for _ in range(100):
    x += 1
torch.cuda.synchronize()
def t(f, n=2000):
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    return (time.perf_counter() - t0) / n * 1e6
print("torch.cuda.synchronize() on an idle device: %.2f us" % t(torch.cuda.synchronize))
ev = torch.cuda.Event()
def rec_poll():
    ev.record()
    while not ev.query():
        pass
print("event record + poll on an idle stream: %.2f us" % t(rec_poll))
def k_then_sync():
    x.add_(1)
    torch.cuda.synchronize()
print("tiny kernel + synchronize: %.2f us" % t(k_then_sync))
def k_then_poll():
    x.add_(1)
    ev.record()
    while not ev.query():
        pass
print("tiny kernel + record + poll: %.2f us" % t(k_then_poll))
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
def k_then_hipsync():
    x.add_(1)
    hip.hipDeviceSynchronize()
print("tiny kernel + hipDeviceSynchronize (ctypes): %.2f us" % t(k_then_hipsync))
s = torch.cuda.current_stream().cuda_stream
def k_then_streamsync():
    x.add_(1)
    hip.hipStreamSynchronize(ctypes.c_void_p(s))
print("tiny kernel + hipStreamSynchronize: %.2f us" % t(k_then_streamsync))
