"""Static checks on the gfx950 assembly of the 4-wave GEMM kernels (gemm_bf16_w4p, and the 8-bit gemm_f8_w4 built the same way): their MFMAs are asm statements, which hides
them from the compiler's hazard recognizer, and three things that went wrong on the GPU because of that are visible in the
instruction stream (docs/experiment_log.md §4, "three traps").  The checks read the disassembly of the object the in-tree build produced (no GPU needed).

  1. no accumulator read (v_accvgpr_read) inside or right behind the MFMA stream of a step;
  2. no 16-byte buffer store whose data registers are rewritten by the next instructions (hipcc does not protect MUBUF
     stores with an SGPR offset);
  3. no s_waitcnt vmcnt(0) and no scratch traffic between the first and the last workgroup barrier of a tile's K loop
     (either one drains the LDS-DMA prefetch ring);
  4. the specialised instantiations with light epilogues use no scratch at all.
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


@pytest.fixture(scope="module")
def w4_kernels(tmp_path_factory):
    """(instruction lists, scratch bytes) of every gemm_bf16_w4p instantiation, disassembled from the object the build made"""
    if not (os.path.exists(OBJDUMP) and os.path.exists(READELF)):
        pytest.skip("llvm-objdump / llvm-readelf not available")
    from multimodaldiscussiontransformer_amd import build as B
    B.build()                                           # no-op when the objects are newer than the sources
    kernels, scratch = {}, {}
    for objname, want in (("gemm.o", "gemm_bf16_w4p"), ("gemm_f8.o", "gemm_f8_w4")):
        tmp = tmp_path_factory.mktemp("isa")
        obj = shutil.copy(os.path.join(B.HERE, "build", objname), tmp / objname)
        subprocess.run([OBJDUMP, "--offloading", str(obj)], cwd=tmp, capture_output=True, text=True, check=True)   # unbundles beside it
        co = [f for f in os.listdir(tmp) if "amdgcn" in f]
        assert len(co) == 1, os.listdir(tmp)
        co = str(tmp / co[0])
        dis = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", co], capture_output=True, text=True, check=True).stdout.split("\n")
        cur = None
        for l in dis:
            m = re.match(r"^[0-9a-f]+ <(\w+)>:", l)
            if m:
                cur = kernels.setdefault(m.group(1), []) if want in m.group(1) else None
                continue
            if cur is not None:
                t = l.split("//")[0].strip()
                if t:
                    cur.append(t)
        notes = subprocess.run([READELF, "--notes", co], capture_output=True, text=True, check=True).stdout
        for blk in re.split(r"\n\s+- \.", notes):
            n = re.search(r"\.name:\s+(\S+)", blk) or re.search(r"^name:\s+(\S+)", blk, re.M)
            ps = re.search(r"private_segment_fixed_size:\s+(\d+)", blk)
            if n and ps:
                scratch[n.group(1)] = int(ps.group(1))
    assert sum("gemm_bf16_w4p" in k for k in kernels) >= 10 and sum("gemm_f8_w4" in k for k in kernels) >= 4, sorted(kernels)
    return kernels, scratch


def _specialised(name):
    m = re.search(r"ILb[01]ELb[01]ELi(\d+)E", name)      # "Lin1E" = -1 is the runtime-flag kernel
    if m is None:
        m = re.search(r"gemm_f8_w4ILi[01]ELi(\d+)ELi\d+E", name)      # <operand format, epilogue, further direct stores>
    return int(m.group(1)) if m else None


def _vregs(tok):
    r = re.match(r"v\[(\d+):(\d+)\]", tok)
    if r:
        return set(range(int(r.group(1)), int(r.group(2)) + 1))
    r = re.match(r"v(\d+)$", tok)
    return {int(r.group(1))} if r else set()


def test_no_accumulator_read_in_the_mfma_stream(w4_kernels):
    kernels, _ = w4_kernels
    for name, body in kernels.items():
        for i, t in enumerate(body):
            if not t.startswith("v_accvgpr_read"):
                continue
            if any(x.startswith("v_mfma") for x in body[i:i + 12]):
                pytest.fail(f"{name}: {t} within the MFMA stream (instruction {i})")
            for k in range(i - 1, max(0, i - 8), -1):            # an MFMA shortly before: the wait states in between
                if body[k].startswith("v_mfma"):
                    ws = sum(int(x.split()[1]) + 1 for x in body[k + 1:i] if x.startswith("s_nop"))
                    assert ws >= 12, f"{name}: {t} only {ws} wait states behind {body[k]}"
                    break


def test_store_data_registers_are_not_rewritten_behind_the_store(w4_kernels):
    kernels, _ = w4_kernels
    for name, body in kernels.items():
        for i, t in enumerate(body):
            m = re.match(r"buffer_store_dwordx4 (v\[\d+:\d+\])", t)
            if not m:
                continue
            data = _vregs(m.group(1))
            for nxt in body[i + 1:i + 3]:
                if nxt.startswith("s_nop"):
                    break
                mm = re.match(r"v_\w+\s+(v\d+|v\[\d+:\d+\])", nxt)
                if mm and _vregs(mm.group(1)) & data and not nxt.startswith(("v_cmp", "v_cmpx")):
                    pytest.fail(f"{name}: '{nxt}' rewrites the data of '{t}' without wait states")


def test_nothing_drains_the_prefetch_ring_inside_the_k_loop(w4_kernels):
    kernels, _ = w4_kernels
    for name, body in kernels.items():
        if _specialised(name) is None:             # the runtime-flag kernel is a fallback, not a hot path
            continue
        bars = [i for i, t in enumerate(body) if t == "s_barrier"]
        assert len(bars) > (4 if "gemm_f8_w4" in name else 10), name
        lo, hi = bars[1], bars[-1]             # bars[0] is the prologue's; the last one opens the tile's last step
        for i in range(lo, hi):
            t = body[i]
            assert not t.startswith("scratch_"), f"{name}: {t} inside the K loop (instruction {i})"
            assert "vmcnt(0)" not in t, f"{name}: {t} inside the K loop (instruction {i})"


def test_light_epilogues_use_no_scratch(w4_kernels):
    _, scratch = w4_kernels
    light = {0, 1, 4, 69}                      # plain, bias, residual, bias + dropout + residual
    seen = 0
    for name, b in scratch.items():
        if "gemm_bf16_w4p" in name and _specialised(name) in light:
            seen += 1
            assert b == 0, f"{name}: {b} bytes of scratch per lane"
    assert seen >= 6
    f8 = {n: b for n, b in scratch.items() if "gemm_f8_w4" in n}
    light8 = {n: b for n, b in f8.items() if _specialised(n) in (0, 1, 4)}      # plain, bias, residual
    assert len(f8) >= 8 and len(light8) >= 3 and all(b == 0 for b in light8.values()), f8


# ----------------------------------------------------------------------------- attn_bwd_v5 (persistent one-pass attention backward)
@pytest.fixture(scope="module")
def v5_kernels(tmp_path_factory):
    if not (os.path.exists(OBJDUMP) and os.path.exists(READELF)):
        pytest.skip("llvm-objdump / llvm-readelf not available")
    from multimodaldiscussiontransformer_amd import build as B
    B.build()
    tmp = tmp_path_factory.mktemp("isa_attn")
    obj = shutil.copy(os.path.join(B.HERE, "build", "attention_v2.o"), tmp / "attention_v2.o")
    subprocess.run([OBJDUMP, "--offloading", str(obj)], cwd=tmp, capture_output=True, text=True, check=True)
    co = [f for f in os.listdir(tmp) if "amdgcn" in f]
    assert len(co) == 1, os.listdir(tmp)
    co = str(tmp / co[0])
    dis = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", co], capture_output=True, text=True, check=True).stdout.split("\n")
    kernels, cur = {}, None
    for l in dis:
        m = re.match(r"^[0-9a-f]+ <(\w+)>:", l)
        if m:
            cur = kernels.setdefault(m.group(1), []) if "attn_bwd_v5_kernel" in m.group(1) else None
            continue
        if cur is not None:
            t = l.split("//")[0].strip()
            if t:
                cur.append(t)
    notes = subprocess.run([READELF, "--notes", co], capture_output=True, text=True, check=True).stdout
    scratch = {}
    for blk in re.split(r"\n\s+- \.", notes):
        n = re.search(r"\.name:\s+(\S+)", blk) or re.search(r"^name:\s+(\S+)", blk, re.M)
        ps = re.search(r"private_segment_fixed_size:\s+(\d+)", blk)
        if n and ps:
            scratch[n.group(1)] = int(ps.group(1))
    assert len(kernels) == 2, sorted(kernels)          # with and without dropout
    return kernels, scratch


def test_persistent_attention_backward_keeps_its_requests_in_flight(v5_kernels):
    """csrc/attention_v2.hip attn_bwd_v5: the next item's rows are requested after the staging barrier and must stay in flight
    through phase 1, its fragments through phase 2.  What broke that while the kernel was written, each visible in the ISA:
    scratch (a reload is a vector-memory instruction and is followed by s_waitcnt vmcnt(0), which also waits for every store
    and request issued before it); vector loads of seq_ids / seq_offsets behind the first store (same wait); a full drain
    before K is written to LDS.  So: no scratch at all, and between the first workgroup barrier of the item loop and the
    end of the kernel no s_waitcnt vmcnt(0) except the ones that wait for the mask bytes (a branch the hot launches skip:
    they follow a global_load_ubyte) and the one that ends the next item's staging (right before the loop's first barrier)."""
    kernels, scratch = v5_kernels
    for name, body in kernels.items():
        assert scratch.get(name, 0) == 0, (name, scratch.get(name))
        assert not any(t.startswith("scratch_") for t in body), name
        bars = [i for i, t in enumerate(body) if t.startswith("s_barrier")]
        assert len(bars) == 4, (name, len(bars))       # staged | dS^T complete | K written | item done
        # the loop body in program order: from the first barrier to the end; the staging of the next item sits before it
        drains = [i for i in range(bars[0], len(body)) if re.match(r"s_waitcnt vmcnt\(0\)", body[i])]
        for i in drains:
            near = body[max(0, i - 4):i]
            assert any(t.startswith("global_load_ubyte") for t in near), (name, i, body[max(0, i - 6):i + 1])
        # index tables by scalar loads only
        assert not any(t.startswith("global_load_dword ") for t in body[bars[1]:bars[2]]), name


# ----------------------------------------------------------------------------- gemm_bf16_pp256 / pp256p (8-wave ping-pong ring)
@pytest.fixture(scope="module")
def pp_kernels(tmp_path_factory):
    if not os.path.exists(OBJDUMP):
        pytest.skip("llvm-objdump not available")
    from multimodaldiscussiontransformer_amd import build as B
    B.build()
    tmp = tmp_path_factory.mktemp("isa_pp")
    obj = shutil.copy(os.path.join(B.HERE, "build", "gemm.o"), tmp / "gemm.o")
    subprocess.run([OBJDUMP, "--offloading", str(obj)], cwd=tmp, capture_output=True, text=True, check=True)
    co = [f for f in os.listdir(tmp) if "amdgcn" in f]
    assert len(co) == 1, os.listdir(tmp)
    dis = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", str(tmp / co[0])], capture_output=True, text=True, check=True).stdout.split("\n")
    kernels, cur = {}, None
    for l in dis:
        m = re.match(r"^[0-9a-f]+ <(\w+)>:", l)
        if m:
            cur = kernels.setdefault(m.group(1), []) if re.search(r"gemm_bf16_pp256p?I", m.group(1)) else None
            continue
        if cur is not None:
            t = l.split("//")[0].strip()
            if t:
                cur.append(t)
    assert sum("pp256pI" in k for k in kernels) >= 20 and sum("pp256I" in k for k in kernels) >= 12, sorted(kernels)
    return kernels


def _mfma_segments(body):
    """[first, last] instruction indices of every run of MFMAs that makes up one 32-k step's MFMA segment (>= 32 of them, only
    s_waitcnt / s_nop in between)"""
    segs, i = [], 0
    while i < len(body):
        if body[i].startswith("v_mfma"):
            j, n, last = i, 0, i
            while j < len(body) and (body[j].startswith(("v_mfma", "s_waitcnt", "s_nop", "s_andn2", "s_and_b64"))):
                if body[j].startswith("v_mfma"):
                    n, last = n + 1, j
                j += 1
            if n >= 32:
                segs.append((i, last))
            i = j
        else:
            i += 1
    return segs


def test_ping_pong_ring_keeps_its_segment_discipline(pp_kernels):
    """The RAW / WAR argument of the ring (csrc/gemm.hip, above gemm_bf16_pp256) holds for the instruction stream only if every
    fragment read and every LDS-DMA request of a 32-k step sits in the step's READ segment, closed by s_waitcnt lgkmcnt(0) and a
    workgroup barrier, and the MFMA segment between its two barriers touches neither LDS nor memory.  The barriers are builtins
    the compiler sees as not touching memory: nothing but the asm waits and the scheduling fences keeps an LDS read from moving
    across one.  Audited here, for every instantiation (production and the jittered stress build):
      * an MFMA segment is entered through  s_waitcnt lgkmcnt(0) ... s_barrier  with no LDS / vector-memory instruction between
        the wait and the first MFMA, and left through a barrier with none between the last MFMA and the barrier;
      * no LDS or vector-memory instruction inside an MFMA segment;
      * the fragment reads that follow the closing barrier come BEFORE the first LDS-DMA request of that READ segment only in
        program order that respects the barrier (i.e. there is no ds_read between the barrier and the MFMAs it opens)."""
    mem = ("ds_", "buffer_", "global_", "flat_", "scratch_")
    audited = 0
    for name, body in pp_kernels.items():
        segs = _mfma_segments(body)
        assert segs, name
        for first, last in segs:
            # inside: only MFMAs, waits (no-ops behind the lgkmcnt(0) in front of the barrier) and nops
            inside = [t for t in body[first:last + 1] if not t.startswith(("v_mfma", "s_waitcnt lgkmcnt", "s_nop", "s_andn2", "s_and_b64"))]
            assert not inside, (name, first, inside[:3])
            # entry: walking back from the first MFMA we must meet s_barrier, then s_waitcnt lgkmcnt(0), before any memory instruction
            k, seen_bar, ok = first - 1, False, False
            while k >= 0 and first - k < 40:
                t = body[k]
                if t.startswith(mem):
                    break
                if t == "s_barrier":
                    seen_bar = True
                elif seen_bar and re.match(r"s_waitcnt.*lgkmcnt\(0\)", t):
                    ok = True
                    break
                elif t.startswith(("s_cbranch", "s_branch")) and not seen_bar:
                    pass                                  # the jitter loops of the stress build sit between barrier and MFMAs
                k -= 1
            assert ok, (name, first, body[max(0, first - 12):first + 1])
            # exit: the next barrier comes before any memory instruction (the fp32-output kernel's riding bias-gradient MFMAs sit
            # in branches of their own behind the main run: a branch out of the linear view ends the walk)
            k = last + 1
            while k < len(body) and body[k] != "s_barrier" and not body[k].startswith("s_branch"):
                assert not body[k].startswith(mem), (name, last, body[last:k + 1])
                k += 1
            assert k < len(body), name
            audited += 1
    assert audited >= 40


def test_first_tile_stand_in_stores_are_all_there(w4_kernels):
    """Round 4's root cause of the intermittent garbage tiles: steps 0-2 of a tile count the previous tile's direct stores among the
    operations in flight; before a workgroup's FIRST tile dropped stores stand in for them — and as sixteen identical builtin stores
    they were merged into one by the compiler, so the first tile's first waits were 15 operations too lax (no wait at all) and its
    first fragments could be read from a stage that had not landed.  They are asm statements now; this counts them: between the
    prologue's barrier and the first step's barrier of every specialised 4-wave instantiation sit exactly as many 16-byte buffer
    stores as the budget of step 0 assumes beyond its 16 LDS-DMA pieces."""
    kernels, _ = w4_kernels
    seen = 0
    for name, body in kernels.items():
        if _specialised(name) is None or "gemm_f8_w4" in name:
            continue
        bars = [i for i, t in enumerate(body) if t == "s_barrier"]
        stores = sum(1 for t in body[bars[0]:bars[1]] if t.startswith("buffer_store_dwordx4"))
        waits = [t for t in body[bars[0]:bars[1]] if t.startswith("s_waitcnt vmcnt")]
        assert waits, name
        budget = int(re.search(r"vmcnt\((\d+)\)", waits[-1]).group(1))          # the wait in front of step 0's barrier
        assert stores == budget - 16, f"{name}: {stores} stand-in stores in front of step 0, whose budget vmcnt({budget}) counts {budget - 16}"
        seen += 1
    assert seen >= 8
