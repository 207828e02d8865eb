"""Instruction mix of the inner loops of one kernel in the -save-temps ISA listing: python exp/loopstat.py <mangled-name-substring>"""
import sys, re
L = open('/tmp/sgmx-hip-amdgcn-amd-amdhsa-gfx950.s').read().split('\n')
name = sys.argv[1]
start = next(i for i, l in enumerate(L) if l.startswith('_ZN') and name in l and ':' in l)
end = next(i for i in range(start, len(L)) if 's_endpgm' in L[i])
for i in range(start, end):
    if 'Inner Loop Header' in L[i]:
        cnt = {}
        j = i
        while j < end:
            l = L[j].strip()
            if l and not l.startswith(';') and not l.startswith('.'):
                op = l.split()[0]
                k = 'valu' if op.startswith('v_') else ('salu' if op.startswith('s_') and op.split('_')[1] not in ('waitcnt', 'nop', 'cbranch') else op)
                if op == 'v_mov_b32_e32': k = 'v_mov'
                cnt[k] = cnt.get(k, 0) + 1
                if op.startswith('s_cbranch_scc') or op.startswith('s_cbranch_vcc'):
                    break
            j += 1
        print(i - start, j - i, cnt)
