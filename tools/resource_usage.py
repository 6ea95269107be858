"""VGPR / scratch / occupancy per kernel out of hipcc -Rpass-analysis=kernel-resource-usage (used by tools/build_exp.sh)."""
import re,sys
log=open(sys.argv[1]).read()
blocks=re.split(r'remark: [^\n]*Function Name: ', log)[1:]
for b in blocks:
    name=b.split('\n')[0].split(' [')[0]
    def g(k):
        m=re.search(k+r': (\d+)', b); return m.group(1) if m else '?'
    short=re.sub(r'_ZN6dctfhe','',name)[:58]
    sc=g(r'ScratchSize \[bytes/lane\]'); oc=g(r'Occupancy \[waves/SIMD\]'); ld=g(r'LDS Size \[bytes/block\]')
    print("%-60s VGPR=%s AGPR=%s scratch=%s occ=%s lds=%s"%(short,g('VGPRs'),g('AGPRs'),sc,oc,ld))
