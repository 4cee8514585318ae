import time, torch, sys
sys.path.insert(0, ".")
from spectrum_vae_amd import ops
dev = torch.device("cuda", 0)
x = torch.zeros(64, device=dev)
big = torch.zeros(1 << 22, device=dev)
def run(name, fn, n=200, reps=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps / n * 1e6
    print(f"{name:40s} {dt:6.2f} us per node")
run("sv fill 64 floats", lambda: ops.fill(x, 1.0))
run("torch add_ 64 floats", lambda: x.add_(1.0))
run("sv fill 4M floats (16 MB)", lambda: ops.fill(big, 1.0))
run("torch add_ 4M floats", lambda: big.add_(1.0))
y = torch.zeros(1 << 18, device=dev)
run("sv fill 256K floats", lambda: ops.fill(y, 1.0))
# eager for comparison
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(2000): ops.fill(x, 1.0)
torch.cuda.synchronize(); print("eager sv fill", (time.perf_counter() - t0) / 2000 * 1e6, "us per launch")
