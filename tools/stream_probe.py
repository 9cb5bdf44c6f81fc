"""Do a pool stream and the default stream overlap on this box?  (diagnostic for pipeline()'s f0 side stream)"""
import time, torch
dev = torch.device("cuda:0")
a = torch.randn(4096, 4096, device=dev)
x = torch.randn(1 << 20, device=dev)


def small_work(n=200):
    y = x
    for _ in range(n):
        y = y * 1.0001 + 0.5
    return y


def run(main_stream, side_stream, label):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(side_stream):
        torch.cuda._sleep(int(2.0e8))  # ~0.1 s of one workgroup spinning
    with torch.cuda.stream(main_stream):
        for _ in range(20):
            b = a @ a
    torch.cuda.synchronize()
    print(label, "%.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)


torch.cuda.synchronize()
t0 = time.perf_counter(); torch.cuda._sleep(int(2.0e8)); torch.cuda.synchronize(); print("sleep alone %.1f ms" % ((time.perf_counter() - t0) * 1e3))
for _ in range(3): b = a @ a
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): b = a @ a
torch.cuda.synchronize(); print("matmuls alone %.1f ms" % ((time.perf_counter() - t0) * 1e3))
run(torch.cuda.default_stream(dev), torch.cuda.Stream(dev), "default + pool :")
run(torch.cuda.Stream(dev), torch.cuda.Stream(dev), "pool + pool    :")
