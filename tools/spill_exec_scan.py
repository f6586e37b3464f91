#!/usr/bin/env python3
"""CPU: scan gfx950 assembly (hipcc -S --cuda-device-only) for VGPR spill code that the register allocator placed at the head of a join block IN FRONT OF the
`s_or_b64 exec, exec, s[..]` that re-activates the lanes -- the stores / reloads then run for the lanes of the branch that just ended only (or none), and every other lane
reads a stale scratch slot later.  Round 6 found exactly this in k_lb_pass<2, OpBlockHeads, 8> when it is compiled under a 128-register bound (tools/lb_bisect.py: wrong and
differently wrong stage-2 block lists, memory faults): see DESIGN.md section 9.  The product build must be clean of the pattern; `make check-spills` runs this over every .hip.
A second, coarser check per spill slot: the nesting depth of exec regions (saveexec opens one, `s_or_b64 exec, exec, ...` closes one) at every store and every reload of the slot --
a slot whose EVERY store sits deeper than some reload is reloaded for lanes nobody stored (it flags the same three slots of that kernel, and nothing in the product).
usage: tools/spill_exec_scan.py file.s [...]   -> exit status 1 if any kernel shows either pattern"""
import re
import sys

bad = 0
for path in sys.argv[1:]:
    kern = None; head = False; pending = []; label = None
    spills = {}
    depth = 0; slots = {}      # second check: slot -> (depths of its stores, depths of its reloads)

    def close_kernel():
        global bad
        for off, (st, ld) in sorted(slots.items()):
            if st and ld and min(ld) < min(st):
                bad += 1
                print(f"{path}: {kern}: spill slot at offset {off}: every store inside exec nesting {sorted(set(st))}, a reload at {sorted(set(ld))}")
    for ln, line in enumerate(open(path), 1):
        s = line.strip()
        m = re.match(r"^(_Z\w+|\w+):\s*(;.*)?$", s)
        if m and not s.startswith(".L"):
            if kern: close_kernel()
            kern = m.group(1); depth = 0; slots = {}
        if re.match(r"s_(and|or|andn2|xor)_saveexec_b64", s): depth += 1
        elif re.match(r"s_or_b64 exec, exec,", s): depth = max(0, depth - 1)
        ms = re.search(r"scratch_(store|load)_dword\w* .*off(?: offset:(\d+))? ; \d+-byte Folded (Spill|Reload)", s)
        if ms:
            st_ld = slots.setdefault(int(ms.group(2) or 0), ([], []))
            (st_ld[0] if ms.group(1) == "store" else st_ld[1]).append(depth)
        if re.match(r"^\.LBB\d+_\d+:", s) or re.match(r"^; %bb\.\d+:", s):      # a block starts at a label or (fall-through) at the "; %bb.N:" marker
            head = True; pending = []; label = s.split(":")[0].lstrip("; "); continue
        if not s or s.startswith(";") or s.startswith("."):
            continue
        if head:
            if re.match(r"s_or_b64 exec, exec, s\[", s):
                if pending:
                    bad += 1
                    print(f"{path}: {kern}: {label}: {len(pending)} spill instruction(s) in front of the exec restore at line {ln}:")
                    for p in pending[:6]: print("     ", p)
                head = False; pending = []; continue
            if "Folded Spill" in s or "Folded Reload" in s:
                pending.append(f"{ln}: {s}")
            # anything else may sit between the block's start and the restore (SGPR spill lanes, scalar code, even vector code of the ending branch's lanes);
            # the block's head ends at the first instruction that writes exec or branches
            if re.search(r"\bexec\b", s.split(",")[0]) or re.match(r"(s_and_saveexec|s_or_saveexec|s_andn2_saveexec|s_cbranch|s_branch|s_endpgm|s_barrier)", s):
                head = False; pending = []
        if "Folded Spill" in s or "Folded Reload" in s:
            spills[kern] = spills.get(kern, 0) + 1
    if kern: close_kernel()
    for k, n in spills.items():
        print(f"{path}: {k}: {n} VGPR spill instructions")
print("pattern found" if bad else "clean")
sys.exit(1 if bad else 0)
