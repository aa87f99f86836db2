"""benchkit -- the parts of bench.py (the driver-contract benchmark at the repo root), one concern per module so that each is testable by itself:

  cpu.py          the cpu_baseline leg: the real reference runtime (oracle/_ref) on the host's cores (the ONLY module here that touches oracle/)
  roofs.py        roofline rows (MFMA / HBM), the pure-MFMA calibration, the PMC traffic digests under profiles/
  attribution.py  per-kernel HIP-event attribution of a net's step -> roofline rows + the per-layer table; the spread over passes
  timing.py       the contract's timed region, one net through the feather::Net runtime, the N > 1 shard check
  convstack.py    --mode convstack: the convolution layers alone through ConvBooster::Forward
  launcher.py     `bench.py --gpus N` without a launcher: start the N ranks (torch.distributed.run)
  report.py       what rank 0 prints: the ONE compact JSON line (<= 8 KiB, enforced) and the detail side file.  No torch, no GPU.
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PEAK_MFMA_F32_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CU x 4 SIMD x 64 FLOP/clk x 2.4 GHz
PEAK_HBM_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E 8 TB/s spec

DEFAULT_BATCH = {"vgg16": 32, "resnet50": 64, "mobilenet_v1": 256, "squeezenet_v1.1": 64}
# sub-batch replicas of the net per GPU (fhip_net_set_sub_batches), measured with tools/dual_stream_bench.py: MobileNet-V1 b256 gains 8 %
# with two (its HBM-bound depthwise kernels run under the other share's MFMA-bound 1x1 kernels), VGG-16 and ResNet-50 gain nothing
SUB_BATCHES = {"mobilenet_v1": 2}
STEADY_STEPS = 200  # length of the cross-check region timed after the contract's K steps ("steady_state" in the detail file)
