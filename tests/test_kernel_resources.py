"""Properties of the compiled gfx950 kernels that no numerical test sees, read from the code objects inside build/obj/*.o
(no GPU needed): the hot kernels must not touch scratch memory, and the kernel-argument prefetch must give every touched
word a scalar register of its own ("the kernel-argument prefetch": one shared destination register let a
late scalar load overwrite a live value — a fault that came and went with register allocation and timing)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def _code_object(tmp_path, name):
    obj = os.path.join(ROOT, "build", "obj", name + ".o")
    if not os.path.exists(obj) or not os.path.exists(os.path.join(LLVM, "llvm-objdump")):
        pytest.skip("no %s or no llvm-objdump in this environment" % obj)
    local = os.path.join(str(tmp_path), name + ".o")
    shutil.copy(obj, local)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], check=True, capture_output=True, cwd=str(tmp_path))
    co = [f for f in os.listdir(str(tmp_path)) if f.startswith(name + ".o.") and "gfx950" in f]
    assert co, "no gfx950 code object inside %s" % obj
    return os.path.join(str(tmp_path), co[0])


def _kernel_notes(co):
    """{kernel symbol: {field: int}} from the code object's metadata note."""
    out = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True, capture_output=True, text=True).stdout
    # one block per kernel, opened by "  - .agpr_count:" (keys are sorted: .group_segment_fixed_size comes BEFORE .name, so the fields
    # are collected per block and attached to the block's name at its end)
    kernels, fields, name = {}, {}, None

    def close():
        if name is not None:
            kernels[name] = dict(fields)
    for line in out.splitlines():
        if re.match(r"\s*- \.agpr_count:", line):
            close()
            fields, name = {}, None
        m = re.match(r"\s*\.name:\s+(\S+)", line)
        if m and m.group(1).startswith("_Z"):
            name = m.group(1)
            continue
        m = re.match(r"\s*(?:- )?\.(private_segment_fixed_size|vgpr_spill_count|sgpr_spill_count|vgpr_count|agpr_count|group_segment_fixed_size):\s+(\d+)", line)
        if m:
            fields[m.group(1)] = int(m.group(2))
    close()
    return kernels


def test_default_16_bit_kernels_use_no_scratch(built, tmp_path):
    k = _kernel_notes(_code_object(tmp_path, "gett_h16"))
    hot = {n: v for n, v in k.items() if "gett_h16_kernel" in n or "gett_h16s_kernel" in n}
    if not hot:
        pytest.skip("production build: the retired eight-wave families are compiled by make RESEARCH=1 only")
    assert len(hot) >= 16, sorted(k)
    bad = {n: v for n, v in hot.items() if v.get("private_segment_fixed_size", 0) or v.get("vgpr_spill_count", 0)}
    assert not bad, bad


def test_lean_four_wave_16_bit_kernels_use_no_scratch_and_all_512_registers(built, tmp_path):
    """gett_h16v.hip: the default 16-bit kernel (gett_h16w4x_kernel, 16x16x32 MFMA) and its 32x32x16 sibling — one wave per SIMD,
    256 accumulator registers + 256 others, no private segment (a scratch allocation is paid for at every dispatch)."""
    k = _kernel_notes(_code_object(tmp_path, "gett_h16v"))
    hot = {n: v for n, v in k.items() if "gett_h16w4x_kernel" in n or "gett_h16w4v_kernel" in n}
    assert len(hot) >= 16, sorted(k)          # 8 layouts x types of the default and their ragged-K twins (+ 8 of the retired 32x32x16 sibling and the measurement-only instantiations in a research build)
    assert sum(1 for n in hot if n.endswith("Lb0ELi0ELb1EEEvNS_10GettParamsE")) == 8, sorted(hot)   # <..., TIMED = false, XST = 0, RAG = true>
    bad = {n: v for n, v in hot.items() if v.get("private_segment_fixed_size", 0) or v.get("vgpr_spill_count", 0)}
    assert not bad, bad
    assert all(v.get("group_segment_fixed_size", 131072) == 131072 for v in hot.values()), hot   # two K-tiles of 64 KiB
    # the 128 x 128 mid-size sibling: no scratch, two K-tiles of 32 KiB, and few enough registers for two workgroups per CU
    mid = {n: v for n, v in k.items() if "gett_h16w4m_kernel" in n}
    assert len(mid) == 32, sorted(k)          # ring of two K-tiles (two workgroups per CU) and of four (one), each with its ragged-K twin
    assert not {n: v for n, v in mid.items() if v.get("private_segment_fixed_size", 0) or v.get("vgpr_spill_count", 0)}, mid
    two = {n: v for n, v in mid.items() if n.endswith("Li2ELb0EEEvNS_10GettParamsE") or n.endswith("Li2ELb1EEEvNS_10GettParamsE")}
    assert len(two) == 16 and all(v.get("group_segment_fixed_size") == 65536 and v.get("vgpr_count", 999) <= 256 for v in two.values()), mid
    assert all(v.get("group_segment_fixed_size") == 131072 for n, v in mid.items() if n not in two), mid
    # the 64 x 64 kernel for small problems: no scratch, four K-tiles of 16 KiB, registers for two workgroups per CU
    small = {n: v for n, v in k.items() if "gett_h16w4q_kernel" in n}
    assert len(small) >= 8, sorted(k)
    assert not {n: v for n, v in small.items() if v.get("private_segment_fixed_size", 0) or v.get("vgpr_spill_count", 0)}, small
    assert all(v.get("group_segment_fixed_size") == 65536 and v.get("vgpr_count", 999) <= 256 for v in small.values()), small


def test_inline_asm_mfma_kernel_waits_before_it_reads_its_accumulators(built, tmp_path):
    """The 16x16x32 kernels issue their MFMAs from inline asm (the compiler's hazard recognizer pads independent 4-pass MFMAs to 27
    cycles), so the compiler does not know that the accumulators are written late: in every instantiation every instruction that
    reads an accumulator register must have the kernel's two `s_nop 15` — not an MFMA — as its nearest predecessor of the two kinds in
    the instruction stream, and no accumulator may be spilled (a spill store right behind an asm MFMA would read the register before
    the matrix pipe has written it)."""
    co = _code_object(tmp_path, "gett_h16v")
    k = _kernel_notes(co)
    names = [n for n in k if any(x in n for x in ("gett_h16w4x_kernel", "gett_h16w4m_kernel", "gett_h16w4q_kernel", "gett_h16w8m_kernel"))]
    assert len(names) >= 40, sorted(k)
    for name in names:
        dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--disassemble-symbols=" + name, co], check=True,
                             capture_output=True, text=True).stdout.splitlines()
        ins = [l.split("//")[0].strip() for l in dis if l.startswith("\t")]
        mfma = [i for i, l in enumerate(ins) if l.startswith("v_mfma_f32_16x16x32")]
        assert len(mfma) >= (3 * 128 if "w4x" in name else 7 * 8 if "w4q" in name else 3 * 32), (name, len(mfma))      # the unrolled K-tiles and the tail
        # (v_accvgpr_mov a, a only spreads the zero the accumulators start from)
        reads = [i for i, l in enumerate(ins) if l.startswith("v_accvgpr_read")
                 or re.match(r"(ds_write|ds_store|global_store|buffer_store|flat_store|scratch_store)\S* .*\ba\[?\d", l)]
        assert reads, name
        for i in reads:
            j = i - 1
            while j >= 0 and not ins[j].startswith("v_mfma") and not ins[j].startswith("s_nop 15"):
                j -= 1
            assert j >= 1 and ins[j].startswith("s_nop 15") and ins[j - 1].startswith("s_nop 15"), (name, i, ins[max(j - 2, 0):i + 1][:12])
        assert not any(l.startswith("scratch_") for l in ins), name


def test_fp32_stream_kernels_do_not_spill_vector_registers(built, tmp_path):
    k = _kernel_notes(_code_object(tmp_path, "gett_f32_stream"))
    hot = {n: v for n, v in k.items() if "gett_f32_stream_kernel" in n}
    assert len(hot) >= 19, sorted(k)
    bad = {n: v for n, v in hot.items() if v.get("vgpr_spill_count", 0)}
    assert not bad, bad
    # the headline einsum's kernel and the contraction.cu default's: no private segment at all
    for cfg in ("Li96ELi96ELi1ELi0ELi3ELi0", "Li96ELi96ELi0ELi0ELi3ELi0", "Li128ELi128ELi1ELi0ELi3ELi0"):
        (name,) = [n for n in hot if cfg in n]
        assert hot[name].get("private_segment_fixed_size", 0) == 0, (name, hot[name])


@pytest.mark.parametrize("obj,symbol", [
    ("gett_f32_stream", "_ZN5ctamd22gett_f32_stream_kernelINS_9StreamCfgILi96ELi96ELi1ELi0ELi3ELi0ELb0EEEEEvNS_10GettParamsE"),
    ("gett_h16v", "_ZN5ctamd18gett_h16w4x_kernelILb1ELi1ELi0ELb0ELi0ELb0EEEvNS_10GettParamsE"),
    ("gett_h16p", "_ZN5ctamd18gett_h16w4p_kernelILb1ELi1ELi0EEEvNS_10GettParamsE"),
])
def test_argument_prefetch_keeps_one_register_per_touched_line(built, tmp_path, obj, symbol):
    co = _code_object(tmp_path, obj)
    dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--disassemble-symbols=" + symbol, co], check=True,
                         capture_output=True, text=True).stdout
    # the burst: s_load_dword sN, s[a:b], 0x40 * i — one per 64-byte line of the 848-byte argument block
    loads = re.findall(r"s_load_dword (s\d+), s\[\d+:\d+\], (0x[0-9a-f]+)", dis)
    # (the FIRST load of each line: the fp32 kernels' epilogue re-reads single fields through a laundered argument pointer much later, and
    # some of those fields sit at line starts)
    burst = {}
    for reg, off in loads:
        if int(off, 16) % 64 == 0 and int(off, 16) < 896:
            burst.setdefault(int(off, 16), reg)
    assert sorted(burst) == [64 * i for i in range(14)], loads[:20]
    assert len(set(burst.values())) == 14, burst          # fourteen lines, fourteen different destination registers


@pytest.mark.parametrize("obj", ["gett_gen_h16", "gett_gen_f64", "gett_gen_cplx"])
def test_general_mfma_family_uses_no_scratch(built, tmp_path, obj):
    """gett_gen_kernel is compiled with __launch_bounds__(256, 2) = 256 registers per lane; the fp64 128 x 128 x 16 tile (128
    accumulator + 64 fragment + 32 staging registers) and the three-accumulator complex tiles are the tight ones.  A spill would be a
    scratch allocation at every dispatch (and lost occupancy): every instantiation must come out with no private segment."""
    k = _kernel_notes(_code_object(tmp_path, obj))
    gen = {n: v for n, v in k.items() if "gett_gen_kernel" in n}
    assert len(gen) >= 8, sorted(k)
    bad = {n: v for n, v in gen.items() if v.get("private_segment_fixed_size", 0) or v.get("vgpr_spill_count", 0)}
    assert not bad, bad
    assert all(v.get("vgpr_count", 999) <= 256 for v in gen.values()), gen      # two workgroups per CU


def test_persistent_16_bit_kernel_uses_no_scratch_and_all_of_the_lds(built, tmp_path):
    """gett_h16p.hip (round 5): the persistent 256 x 256 kernel keeps the next tile's staging tables and odometer alive during its
    epilogue — still no private segment, no spilled vector register, 256 accumulator registers, and exactly 160 KiB of LDS (the 128-KiB
    ring + two 4-KiB epilogue images per wave); its accumulator reads wait behind the two s_nop 15 like every inline-asm MFMA kernel."""
    co = _code_object(tmp_path, "gett_h16p")
    k = _kernel_notes(co)
    hot = {n: v for n, v in k.items() if "gett_h16w4p_kernel" in n}   # 2 types x 4 operand layouts (the measurement instantiations went in round 6)
    assert len(hot) == 8, sorted(k)
    bad = {n: v for n, v in hot.items() if v.get("private_segment_fixed_size", 0) or v.get("vgpr_spill_count", 0)}
    assert not bad, bad
    assert all(v.get("group_segment_fixed_size") == 163840 and v.get("agpr_count") == 256 for v in hot.values()), hot
    for name in hot:
        dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--disassemble-symbols=" + name, co], check=True,
                             capture_output=True, text=True).stdout.splitlines()
        ins = [l.split("//")[0].strip() for l in dis if l.startswith("\t")]
        assert sum(1 for l in ins if l.startswith("v_mfma_f32_16x16x32")) >= 3 * 128, name
        reads = [i for i, l in enumerate(ins) if l.startswith("v_accvgpr_read")
                 or re.match(r"(ds_write|ds_store|global_store|buffer_store|flat_store|scratch_store)\S* .*\ba\[?\d", l)]
        assert reads, name
        for i in reads:
            j = i - 1
            while j >= 0 and not ins[j].startswith("v_mfma") and not ins[j].startswith("s_nop 15"):
                j -= 1
            assert j >= 1 and ins[j].startswith("s_nop 15") and ins[j - 1].startswith("s_nop 15"), (name, i, ins[max(j - 2, 0):i + 1][:12])
        assert not any(l.startswith("scratch_") for l in ins), name
        # the transposed epilogue: transposing reads outside the main loop's count, 8-byte LDS writes, global (not flat) 16-byte stores
        assert any(l.startswith("ds_write2st64_b64") or l.startswith("ds_write_b64") for l in ins), name
        assert sum(1 for l in ins if l.startswith("ds_write_b128")) >= 32 + 32 + 32, name      # the row image: 4 parks x 8 passes, beta == 0 and beta != 0, + 4 chunks of C x 8 passes
        assert sum(1 for l in ins if l.startswith("ds_read_b64_tr_b16")) >= 2 * 64 + 64, name       # transposing reads of the epilogue: 8 per pass, both forms, + 8 fragments of C per pass
        assert sum(1 for l in ins if l.startswith("global_store_dwordx4")) >= 32, name
