#!/usr/bin/env python3
"""Register use of the kernels of libsgmx.so from `make -C stereo_vision_amd/csrc asm` (/tmp/sgmx_resource.txt).
usage: exp/resusage.py [substring of the mangled name]"""
import re, sys
pat = sys.argv[1] if len(sys.argv) > 1 else "aggregate3_kernel"
t = open('/tmp/sgmx_resource.txt').read()
for b in re.split(r'Function Name: ', t)[1:]:
    name = b.split(' ')[0]
    if pat not in name:
        continue
    f = lambda k: re.search(k + r': (\d+)', b).group(1)
    print("%-90s VGPR %3s spill %2s scratch %3s occ %s" % (name[9:99], f('VGPRs'), f('VGPRs Spill'), f(r'ScratchSize \[bytes/lane\]'), f(r'Occupancy \[waves/SIMD\]')))
