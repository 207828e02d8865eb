"""Instruction mix of one steady-state step (unmasked, no edge stores) of aggregate3_kernel<RP=4> from the
compiler's assembly (make -C stereo_vision_amd/csrc asm): counts per opcode over the PF-step loop body / PF."""
import collections, re, sys
S = sys.argv[1] if len(sys.argv) > 1 else "/tmp/sgmx-hip-amdgcn-amd-amdhsa-gfx950.s"
name = sys.argv[2] if len(sys.argv) > 2 else "_ZN4sgmx17aggregate3_kernelILi4ELb0ELb0ELi4ELi0EEEvPKjS2_PhPtS4_NS_8Agg2ArgsE"
PF = int(sys.argv[3]) if len(sys.argv) > 3 else 4
lines = open(S).read().split(name + ":")[1].split("s_endpgm")[0].splitlines()
# the first inner loop (Depth=2) after the dispatcher is the unmasked / no-edge one: the loop whose body has no cndmask
loops, cur = [], None
for l in lines:
    if re.match(r"\.LBB\d+_\d+:", l):
        if cur is not None:
            loops.append(cur)
        cur = [l]
    elif cur is not None:
        cur.append(l)
best = None
for i, blk in enumerate(loops):
    if "Inner Loop Header" in "".join(blk[:4]):
        body = blk
        k = i + 1
        while k < len(loops) and "s_cbranch" not in "".join(body):
            body = body + loops[k]; k += 1
        ops = [x.split()[0] for x in body if re.match(r"\s+[vs]_|\s+buffer_|\s+ds_", x)]
        if "v_cndmask_b32_e32" not in ops and "v_bcnt_u32_b32" in ops and best is None:
            best = ops
c = collections.Counter(best)
valu = {k: v for k, v in c.items() if k.startswith("v_")}
fast = sum(v for k, v in valu.items() if k.split("_e")[0] in ("v_xor_b32", "v_not_b32", "v_and_b32", "v_or_b32", "v_add_u32", "v_sub_u32", "v_mov_b32"))
tot = sum(valu.values())
print("per step (loop body / %d): VALU %.2f of which full-rate class %.2f (share %.3f); SALU %.2f; VMEM %.2f" % (
    PF, tot / PF, fast / PF, fast / tot, sum(v for k, v in c.items() if k.startswith("s_")) / PF,
    sum(v for k, v in c.items() if k.startswith("buffer_")) / PF))
for k, v in sorted(valu.items(), key=lambda kv: -kv[1]):
    print("  %-22s %.2f" % (k, v / PF))
