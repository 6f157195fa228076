"""The hand-counted `s_waitcnt vmcnt(N)` kernels against tools/audit_asm_loads.py (no GPU: it reads the built objects).

The streaming GEMV issues its loads from inline asm; the compiler does not know their destinations are in flight.
A build whose register allocation reads one of them before the wait is wrong on the GPU only some of the time --
this catches the straight-line form of it at build time."""
import glob
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("audit_asm_loads", os.path.join(ROOT, "tools", "audit_asm_loads.py"))
audit_asm_loads = importlib.util.module_from_spec(spec)
spec.loader.exec_module(audit_asm_loads)

OBJ_DIR = os.path.join(ROOT, "dash-infer_amd", "lib", "obj")
# + the K-slice GEMM (round 4: its activation fragments are asm loads behind one explicit wait)
OBJECTS = sorted(glob.glob(os.path.join(OBJ_DIR, "gemv_stream_inst_*.o")) + glob.glob(os.path.join(OBJ_DIR, "gemm_kslice_inst_*.o")))


def test_the_tool_sees_a_read_before_the_wait_and_not_after():
    early = [("global_load_dwordx4", "v[10:13], v1, s[2:3]"), ("global_load_dword", "v20, v1, s[4:5]"),
             ("v_mov_b32_e32", "v30, v11"),            # copy of a register still in flight
             ("s_waitcnt", "vmcnt(1)"),
             ("v_add_f32_e32", "v31, v10, v12"),        # the x4 load has landed (one newer load may be outstanding)
             ("v_mov_b32_e32", "v32, v20"),             # the newer one has not
             ("s_waitcnt", "vmcnt(0)"), ("v_mov_b32_e32", "v33, v20")]
    got = audit_asm_loads.audit_kernel("k", {"entry": early}, ["entry"])
    assert sorted({f.split(":")[0] for f in got}) == ["k entry+2", "k entry+5"], got  # (both rules see the first one)
    # the zero-filling arm of `valid ? load : 0` is another path; a newer load may take the same destination
    arms = {"entry": [("s_cbranch_scc1", "L1")],
            "L0": [("global_load_dwordx4", "v[10:13], v1, s[2:3]"), ("global_load_dwordx4", "v[10:13], v1, s[2:3]"), ("s_branch", "L2")],
            "L1": [("v_mov_b32_e32", "v10, 0"), ("v_mov_b32_e32", "v11, v10")],
            "L2": [("s_waitcnt", "vmcnt(0)"), ("v_mov_b32_e32", "v5, v12")]}
    assert not audit_asm_loads.audit_kernel("k", arms, ["entry", "L0", "L1", "L2"])


def test_the_tool_follows_an_early_activation_load_across_blocks():
    """the bug it was written for: batch 0 of the activations is loaded ahead of the ring fill, and the compiler resolved a
    phi on those registers with copies placed before the wait (row 0 of a batch read before it had landed)"""
    def kernel(copy_before_wait):
        tail = [("v_mov_b64_e32", "v[84:85], v[56:57]"), ("s_waitcnt", "vmcnt(4)")]
        return {"entry": [("global_load_dwordx4", "v[54:57], v3, s[40:41]"), ("s_cbranch_scc1", "L1")],
                "L0": [("global_load_dwordx4", "v[10:13], v75, s[26:27] nt"), ("global_load_dword", "v2, v74, s[24:25]"), ("s_branch", "L2")],
                "L1": [("global_load_dword", "v2, v1, s[18:19]"), ("global_load_dword", "v2, v1, s[18:19]")],
                "L2": [("global_load_dwordx4", "v[14:17], v75, s[26:27] nt"), ("global_load_dword", "v3, v74, s[24:25]")],
                "L3": tail if copy_before_wait else tail[::-1]}, ["entry", "L0", "L1", "L2", "L3"]
    got = audit_asm_loads.audit_kernel("k", *kernel(True))
    assert len(got) == 1 and "L3+0" in next(iter(got)) and "issued in entry" in next(iter(got)), got
    assert not audit_asm_loads.audit_kernel("k", *kernel(False))
    # a wait that leaves more operations outstanding than were issued since does not cover the load
    blocks, order = kernel(False)
    blocks["L3"] = [("s_waitcnt", "vmcnt(5)"), ("v_mov_b64_e32", "v[84:85], v[56:57]")]
    assert len(audit_asm_loads.audit_kernel("k", blocks, order)) == 1


def test_the_tool_sees_a_load_landing_on_a_register_given_to_another_value():
    """the tail's dummy loads went into a scratch variable: a dead definition, so the compiler reused the register (for the
    zero the accumulators are reset from) while the load was still on its way"""
    blocks = {"entry": [("global_load_dword", "v51, v50, s[40:41]"), ("global_load_dword", "v76, v50, s[40:41]"), ("s_branch", "L1")],
              "L1": [("s_waitcnt", "vmcnt(14)"), ("v_mov_b32_e32", "v51, v50"), ("v_mov_b64_e32", "v[44:45], v[50:51]"), ("s_waitcnt", "vmcnt(0)")]}
    got = audit_asm_loads.audit_kernel("k", blocks, ["entry", "L1"])
    assert len(got) == 1 and "L1+1" in next(iter(got)) and "yet to land" in next(iter(got)), got
    # into the slot's own registers, untouched until the wait: fine
    blocks["entry"][0] = ("global_load_dwordx4", "v[2:5], v50, s[40:41] nt")
    assert not audit_asm_loads.audit_kernel("k", blocks, ["entry", "L1"])


@pytest.mark.skipif(not all(os.path.exists(o) for o in OBJECTS) or not os.path.exists(audit_asm_loads.OBJDUMP),
                    reason="objects not built (python -c 'import __graft_entry__ as g; g.build()')")
@pytest.mark.parametrize("obj", OBJECTS, ids=[os.path.basename(o) for o in OBJECTS])
def test_no_register_is_read_while_its_load_is_in_flight(obj):
    kernels, findings = audit_asm_loads.audit(obj, skip=audit_asm_loads.FUSED_KERNELS)
    assert kernels > 0
    assert not findings, "\n".join(findings[:20])


# ---- MFMA results read too early (tools/audit_mfma_hazard.py) -----------------------------------------------------------------
spec2 = importlib.util.spec_from_file_location("audit_mfma_hazard", os.path.join(ROOT, "tools", "audit_mfma_hazard.py"))
audit_mfma_hazard = importlib.util.module_from_spec(spec2)
spec2.loader.exec_module(audit_mfma_hazard)


def test_the_mfma_audit_flags_a_store_right_behind_a_loop_of_mfmas(tmp_path):
    """the form hipcc produced for moe_router_gate_kernel: the loop ends in the MFMA, the ds_write of the accumulator follows the
    loop exit after two scalar instructions -- and the form with the wait states the kernel now carries"""
    body = ".LBB0_1:\n\tglobal_load_dwordx4 v[8:11], v[4:5], off\n\tv_mfma_f32_16x16x32_bf16 v[0:3], v[8:11], v[12:15], v[0:3]\n\ts_cbranch_scc0 .LBB0_1\n"
    bad = tmp_path / "bad.s"
    bad.write_text("k:\n" + body + "\ts_and_b64 vcc, exec, s[20:21]\n\tds_write_b128 v76, v[0:3]\n\ts_endpgm\n")
    good = tmp_path / "good.s"
    good.write_text("k:\n" + body + "\ts_nop 15\n\ts_nop 7\n\ts_and_b64 vcc, exec, s[20:21]\n\tds_write_b128 v76, v[0:3]\n\ts_endpgm\n")
    assert len(audit_mfma_hazard.audit(str(bad))) == 1
    assert audit_mfma_hazard.audit(str(good)) == []
    # chained accumulation and a VALU read after hipcc's own s_nop are fine
    chain = tmp_path / "chain.s"
    chain.write_text("k:\n\tv_mfma_f32_16x16x32_bf16 v[0:3], v[8:11], v[12:15], v[0:3]\n\tv_mfma_f32_16x16x32_bf16 v[0:3], v[8:11], v[16:19], v[0:3]\n"
                     "\ts_nop 6\n\tv_add_f32_e32 v20, v0, v1\n\ts_endpgm\n")
    assert audit_mfma_hazard.audit(str(chain)) == []


@pytest.mark.parametrize("src", ["moe.hip", "gemv_batch_inst_w4_g1.hip"])
def test_no_kernel_reads_an_mfma_result_without_wait_states(src, tmp_path):
    """compiles the file to device assembly (no GPU needed) and audits it: MoE kernels (where the bug was) and the small-batch
    GEMM whose waves combine MFMA tiles through LDS"""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    csrc = os.path.join(ROOT, "dash-infer_amd", "csrc")
    out = tmp_path / (src + ".s")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + csrc,
                    "-fno-gpu-rdc", "-ffp-contract=off", "-S", "--cuda-device-only", "-o", str(out), os.path.join(csrc, src)],
                   check=True, capture_output=True, timeout=600)
    found = audit_mfma_hazard.audit(str(out))
    assert not found, found[:3]
