#!/usr/bin/env python3
"""tools/check_stream_isa.py -- build-time proof obligation of stream_gemm.h on the SHIPPED binary.

The register-streamed 1x1 GEMM keeps a ring of global loads in flight with inline asm (`global_load_* %0` with the destination as a
plain "=v" output) and counted waits (`s_waitcnt vmcnt(N)` whose asm names the ring registers "+v").  hipcc believes a destination is
written the moment the load asm returns, so nothing in the language stops it from copying, spilling or re-using such a register before
the matching wait -- a copy would read the register before the data has landed and the kernel would be silently wrong.  Whether it did
is a property of the generated code, so this tool checks the generated code: it pulls the gfx950 code objects out of
feathercnn_amd/libfeather_hip.so, disassembles every `stream_gemm_kernel` instantiation and replays its instruction stream with a model
of the vmcnt queue (VMEM loads AND stores enter in program order and retire in order; `s_waitcnt vmcnt(N)` retires all but the newest N):

  * no instruction may read or write a VGPR that is the destination of a load still in the queue;
  * no scratch (spill) instruction may appear in these kernels at all;
  * every loop (backward branch) is replayed a second time with the queue it ends with, so the steady state is covered.

Exit code 0 and a one-line summary per kernel when the obligation holds; 1 and the offending instructions otherwise.
tests/test_boundary.py runs it on CPU (no GPU needed); re-run after any ROCm bump or change to stream_gemm.h.
"""
from __future__ import annotations

import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(so_path: str, arch: str = "gfx950"):
    """The device ELF images for `arch` inside a HIP fat binary (uncompressed clang offload bundles)."""
    data = open(so_path, "rb").read()
    out = []
    for m in re.finditer(re.escape(MAGIC), data):
        base = m.start()
        (n,) = struct.unpack_from("<Q", data, base + 24)
        pos = base + 32
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", data, pos)
            triple = data[pos + 24:pos + 24 + tlen].decode()
            pos += 24 + tlen
            if arch in triple and size:
                out.append(data[base + off:base + off + size])
    return out


REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")


def regs_of(text: str):
    s = set()
    for m in REG.finditer(text):
        if m.group(1):
            s.add((m.group(1), int(m.group(2))))
        else:
            s.update((m.group(3), i) for i in range(int(m.group(4)), int(m.group(5)) + 1))
    return s


def parse(disasm: str):
    """-> {function: [(address, mnemonic, operand text, branch target address | None)]}"""
    funcs, cur = {}, None
    for line in disasm.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
        if m:
            cur = m.group(1)
            funcs[cur] = []
            continue
        if cur is None:
            continue
        body, _, comment = line.partition("//")
        body = body.strip()
        if not body or body.startswith("."):
            continue
        am = re.match(r"\s*([0-9A-Fa-f]+):", comment)
        addr = int(am.group(1), 16) if am else None
        tm = re.search(r"<[^>]+\+0x([0-9a-fA-F]+)>", comment)
        parts = body.split(None, 1)
        funcs[cur].append([addr, parts[0], parts[1] if len(parts) > 1 else "", int(tm.group(1), 16) if tm else None])
    for ins in funcs.values():  # branch targets are printed as offsets from the function's first instruction
        if ins and ins[0][0] is not None:
            base = ins[0][0]
            for i in ins:
                if i[3] is not None:
                    i[3] += base
    return funcs


def check_function(name: str, ins):
    """Replay with the vmcnt queue model.  -> (violations, stats)"""
    index = {a: i for i, (a, _, _, _) in enumerate(ins) if a is not None}
    queue = []  # in program order: set of destination registers (empty for stores)
    violations, stats = [], {"loads": 0, "stores": 0, "waits": 0, "max_in_flight": 0, "loops_replayed": 0}

    def step(i, replay):
        _, mn, ops, _ = ins[i]
        if mn.startswith("scratch_"):
            violations.append(f"{name}: spill instruction `{mn} {ops}`")
        pending = set().union(*queue) if queue else set()
        if mn == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", ops)
            if m:
                keep = int(m.group(1))
                while len(queue) > keep:
                    queue.pop(0)
                if not replay:
                    stats["waits"] += 1
            return
        hit = regs_of(ops) & pending
        if hit:
            violations.append(f"{name}: `{mn} {ops}` touches {sorted(hit)[:6]} while a global load into it is still in flight"
                              + (" (loop steady state)" if replay else ""))
        if mn.startswith(("global_load", "buffer_load", "flat_load")):
            queue.append(regs_of(ops.split(",")[0]))
            if not replay:
                stats["loads"] += 1
        elif mn.startswith(("global_store", "buffer_store", "flat_store", "global_atomic")):
            queue.append(set())
            if not replay:
                stats["stores"] += 1
        stats["max_in_flight"] = max(stats["max_in_flight"], len(queue))

    for i, (addr, mn, ops, tgt) in enumerate(ins):
        step(i, False)
        if (mn.startswith("s_cbranch") or mn == "s_branch") and tgt is not None and addr is not None and tgt <= addr and tgt in index:
            stats["loops_replayed"] += 1  # backward branch: one more trip through the loop body with the queue as it stands
            for j in range(index[tgt], i + 1):
                step(j, True)
    return violations, stats


def main(so_path: str) -> int:
    objs = code_objects(so_path)
    if not objs:
        print("no gfx950 code object found in", so_path)
        return 1
    checked, bad = 0, []
    with tempfile.TemporaryDirectory() as d:
        for k, img in enumerate(objs):
            p = os.path.join(d, f"co{k}.elf")
            open(p, "wb").write(img)
            dis = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", p], capture_output=True, text=True).stdout
            for fn, ins in parse(dis).items():
                if "stream_gemm_kernel" not in fn:
                    continue
                v, st = check_function(fn, ins)
                checked += 1
                bad += v
                demangled = subprocess.run(["c++filt", fn], capture_output=True, text=True).stdout.strip()
                print(f"{'FAIL' if v else 'ok  '} {demangled[:80]:80s} loads {st['loads']:3d} stores {st['stores']:3d} waits {st['waits']:3d} "
                      f"max in flight {st['max_in_flight']:3d} loops {st['loops_replayed']}")
    for v in bad[:40]:
        print("  ", v)
    if not checked:
        print("no stream_gemm_kernel found in", so_path)
        return 1
    print(f"{checked} stream_gemm_kernel instantiations checked, {len(bad)} violations")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "feathercnn_amd", "libfeather_hip.so")))
