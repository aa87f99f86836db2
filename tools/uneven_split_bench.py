import sys, time
import numpy as np, torch
from feathercnn_amd import model_zoo
from feathercnn_amd.net import Net
name = sys.argv[1]; splits = [int(v) for v in sys.argv[2].split(",")]
p, b, i, o = model_zoo.MODELS[name]()
rng = np.random.default_rng(0)
def build(n):
    net = Net(fusion=3, graph=True, tuned=True, concurrency=True)
    net.LoadParam(p); net.LoadWeights(b)
    net.FeedInput(i, torch.from_numpy(rng.uniform(-1, 1, (n, 3, 224, 224)).astype(np.float32)).cuda())
    for _ in range(3): net.Forward()
    torch.cuda.synchronize(); return net
nets = [build(n) for n in splits]
for rep in range(2):
    for _ in range(5):
        for n in nets: n.Forward()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        for n in nets: n.Forward()
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 50
    print(name, splits, f"{t*1e3:.3f} ms  {sum(splits)/t:.0f} img/s")
