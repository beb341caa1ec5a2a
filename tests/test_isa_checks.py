"""Build-time ISA checks of the hand-scheduled encoder GEMM (csrc/gemm_pp.hip).  Its fragment loads are `ds_read_b128` inline-asm
statements whose results become valid only at a later, separate `s_waitcnt lgkmcnt(N)` statement; to the compiler the destination
registers are defined at the read, so it could legally copy or spill them between the two and capture stale data (advisor, round
3).  Correctness therefore depends on the generated code, and this test reads it: for every `gemm_pp_kernel` instantiation of the
product build

  * no scratch memory (a spill of a fragment register inside the stage loop is exactly the failure mode), and
  * between a `ds_read_b128` and the `s_waitcnt lgkmcnt` that releases it, no instruction touches the read's destination registers.

A toolchain bump that breaks either fails here, before any GPU sees the kernel.  No GPU needed (hipcc cross-compiles)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REG = re.compile(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b")


def _regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1):
            out.update((m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.add((m.group(4), int(m.group(5))))
    return out


def _kernels(asm):
    lines = asm.split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\S*gemm_pp_kernel\S*:", l)]
    for n, i in enumerate(starts):
        j = next((k for k in range(i, len(lines)) if ".amdhsa_kernel" in lines[k]), len(lines))
        end = next((k for k in range(j, len(lines)) if ".end_amdhsa_kernel" in lines[k]), len(lines))
        yield lines[i].split(":")[0], lines[i:j], lines[j:end]


@pytest.fixture(scope="module")
def gemm_pp_asm(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not found")
    from embodied_captioning_amd.build import FLAGS
    out = tmp_path_factory.mktemp("isa") / "gemm_pp.s"
    flags = [f for f in FLAGS if f != "-fPIC"]
    r = subprocess.run([hipcc, *flags, "-S", "--cuda-device-only", os.path.join(ROOT, "embodied_captioning_amd", "csrc", "gemm_pp.hip"),
                        "-o", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return out.read_text()


def test_gemm_pp_kernels_use_no_scratch(gemm_pp_asm):
    n = 0
    for name, body, meta in _kernels(gemm_pp_asm):
        n += 1
        seg = [l for l in meta if "private_segment_fixed_size" in l]
        assert seg and all(l.split()[-1] == "0" for l in seg), (name, seg)
        assert not any(re.search(r"\bscratch_(load|store)", l) for l in body), name
    assert n >= 8, n            # G8 and bf16, fp32 / operand-type / KV16 outputs, full and half tiles


def test_gemm_pp_fragment_registers_are_untouched_between_read_and_wait(gemm_pp_asm):
    checked = 0
    for name, body, _ in _kernels(gemm_pp_asm):
        pending = []                                    # destination register sets of LDS reads in flight, oldest first
        for l in body:
            t = l.strip()
            if not t or t.startswith((";", ".")):
                continue
            if t.startswith(("s_branch", "s_setpc", "s_endpgm")):
                pending = []                            # nothing falls through: the next line starts from another path
                continue
            if re.match(r"^[.\w$]+:", t) or t.startswith("s_cbranch"):
                continue                                # a label / conditional branch: the fall-through path keeps its reads in flight
            op = t.split()[0]
            if op.startswith("ds_read") or op.startswith("ds_load"):
                dst = _regs(t.split(",")[0])
                for p in pending:
                    assert not (p & dst), (name, "a read overwrites fragments still in flight", t)
                pending.append(dst)
                checked += 1
                continue
            if op == "s_waitcnt":
                m = re.search(r"lgkmcnt\((\d+)\)", t)
                if m:
                    keep = int(m.group(1))
                    pending = pending[len(pending) - keep:] if keep else []
                continue
            # every other instruction - LDS stores and LDS-DMA included: their address and data operands are VGPRs too - must
            # stay off the registers of reads still in flight
            used = _regs(t)
            for p in pending:
                hit = p & used
                assert not hit, (name, "touches a fragment register before its wait", t, sorted(hit)[:4])
            if op.startswith(("ds_write", "ds_store", "ds_bpermute", "ds_swizzle")):
                pending.append(set())                   # counts in lgkmcnt like a read (in order), holds no register of ours
    assert checked > 100, checked
