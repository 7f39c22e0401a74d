"""Static count of SGPR-spill lane moves (v_writelane / v_readlane on VGPRs that no other instruction touches) of one kernel, split by
loop depth.  usage: lanespill.py file.s 'plan_bits_seq_kernelILi1ELi2ELi0E'"""
import re, sys, collections
lines=open(sys.argv[1]).read().split('\n')
key=sys.argv[2]
start=next(i for i,l in enumerate(lines) if l.startswith('_ZN') and key in l and l.split(';')[0].strip().endswith(':'))
end=next(i for i in range(start,len(lines)) if lines[i].startswith('.Lfunc_end'))
body=lines[start+1:end]
ins=[]  # (label-block, text)
blk='entry'; depth=0
blocks=collections.OrderedDict(); blocks[blk]={'depth':0,'ins':[]}
for l in body:
    t=l.split(';')[0].strip()
    if l.startswith('.LBB'):
        blk=l.split(':')[0]
        m=re.search(r'Depth=(\d+)', l)
        blocks[blk]={'depth':None,'ins':[], 'hdr':l}
        continue
    if not t or t.startswith('.'): continue
    blocks[blk]['ins'].append(t)
# loop depth per block from LLVM's comments: "=>This Loop Header: Depth=N" / "Parent Loop BB.. Depth=N" / "in Loop: Header=BB Depth=N"
hdrs=[l for l in body if l.startswith('.LBB')]
cur=0
for name,b in blocks.items():
    h=b.get('hdr','')
    ds=[int(x) for x in re.findall(r'Depth=(\d+)', h)]
    b['depth']=max(ds) if ds else 0
# the comment may continue on following lines (";   Parent Loop ..."): scan body again
name=None
for l in body:
    if l.startswith('.LBB'): name=l.split(':')[0]
    elif name and l.lstrip().startswith(';') and 'Depth=' in l:
        ds=[int(x) for x in re.findall(r'Depth=(\d+)', l)]
        blocks[name]['depth']=max(blocks[name]['depth'], max(ds))
    elif l.strip() and not l.lstrip().startswith(';'): name=None if not l.startswith('.LBB') else name
allins=[t for b in blocks.values() for t in b['ins']]
use=collections.defaultdict(set)
for t in allins:
    op=t.split()[0]
    for v in re.findall(r'\bv(\d+)\b', t): use[int(v)].add(op)
    for a,b_ in re.findall(r'v\[(\d+):(\d+)\]', t):
        for v in range(int(a),int(b_)+1): use[v].add(op)
spillv={v for v,ops in use.items() if ops and ops<= {'v_writelane_b32','v_readlane_b32'}}
print('spill-holder VGPRs:', sorted(spillv))
tot=collections.Counter(); val=collections.Counter(); sal=collections.Counter(); other=collections.Counter()
for b in blocks.values():
    d=b['depth']
    for t in b['ins']:
        op=t.split()[0]
        if op in ('v_writelane_b32','v_readlane_b32'):
            vs=[int(x) for x in re.findall(r'\bv(\d+)\b', t)]
            if vs and vs[0 if op=='v_writelane_b32' else -1] in spillv or any(v in spillv for v in vs):
                tot[(d,op)]+=1
        if op.startswith('v_'): val[d]+=1
        elif op.startswith('s_'): sal[d]+=1
        else: other[d]+=1
for d in sorted(set(list(val)+list(sal))):
    print(f"loop depth {d}: VALU {val[d]:6d} SALU {sal[d]:6d} other {other[d]:5d} | spill writelane {tot[(d,'v_writelane_b32')]:4d} readlane {tot[(d,'v_readlane_b32')]:4d}")
