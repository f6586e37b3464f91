"""CPU: checks on the compiled product that need no GPU.

tools/spill_exec_scan.py looks through the device assembly of every kernel file for VGPR spill code that the register allocator put at the head of a join block IN FRONT OF the
`s_or_b64 exec, exec, ...` that re-activates the lanes (round 6: k_lb_pass<2, OpBlockHeads, 8> under a 128-register bound stored its wave-scan values for lane 63 only and every other
lane read a stale scratch slot -- wrong block lists and memory faults).  The product build must not contain the pattern."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCAN = os.path.join(ROOT, "tools", "spill_exec_scan.py")

BAD = """
_Z6kernelv:
; %bb.0:
	s_and_saveexec_b64 s[0:1], s[2:3]
	s_cbranch_execz .LBB0_2
; %bb.1:
	ds_write_b32 v1, v0
.LBB0_2:
	v_writelane_b32 v127, s16, 23
	scratch_store_dword off, v10, off offset:164 ; 4-byte Folded Spill
	s_or_b64 exec, exec, s[0:1]
	s_barrier
	scratch_load_dword v60, off, off offset:164 ; 4-byte Folded Reload
	s_endpgm
"""
GOOD = BAD.replace("\tscratch_store_dword off, v10, off offset:164 ; 4-byte Folded Spill\n\ts_or_b64 exec, exec, s[0:1]\n",
                   "\ts_or_b64 exec, exec, s[0:1]\n\tscratch_store_dword off, v10, off offset:164 ; 4-byte Folded Spill\n")


def test_scanner_sees_a_spill_in_front_of_the_exec_restore(tmp_path):
    bad = tmp_path / "bad.s"; bad.write_text(BAD)
    good = tmp_path / "good.s"; good.write_text(GOOD)
    r = subprocess.run([sys.executable, SCAN, str(bad)], capture_output=True, text=True)
    assert r.returncode == 1 and "1 spill instruction(s) in front of the exec restore" in r.stdout, r.stdout
    r = subprocess.run([sys.executable, SCAN, str(good)], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip().endswith("clean"), r.stdout


def test_product_kernels_have_no_spill_in_front_of_an_exec_restore():
    r = subprocess.run(["make", "-j8", "check-spills"], cwd=os.path.join(ROOT, "gsalign_amd", "csrc"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.strip().endswith("clean"), (r.stdout[-3000:], r.stderr[-2000:])
