"""Per-kernel register / scratch metadata from a --save-temps .s file."""
import re, subprocess, sys
s=open(sys.argv[1]).read()
md=s[s.index('amdhsa.kernels'):]
ks=md.split('  - .agpr_count')
flt=sys.argv[2] if len(sys.argv)>2 else ''
for k in ks[1:]:
    name=re.search(r'\.name:\s+(\S+)',k).group(1)
    g=lambda key: int(re.search(r'\.'+key+r':\s+(\d+)',k).group(1))
    dn=subprocess.run(['c++filt',name],capture_output=True,text=True).stdout.strip()
    dn=re.sub(r'\(.*','',dn.replace('(anonymous namespace)::','').replace('void fpe::',''))
    if flt and flt not in dn: continue
    print(f"{dn[:80]:80s} vgpr {g('vgpr_count'):4d} sgpr {g('sgpr_count'):4d} scratch {g('private_segment_fixed_size'):5d} vspill {g('vgpr_spill_count'):4d} sspill {g('sgpr_spill_count'):4d} lds {g('group_segment_fixed_size')}")
