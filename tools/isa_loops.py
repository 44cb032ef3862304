import re,sys
lines=open(sys.argv[1]).read().split('\n')
pat=sys.argv[2]
start=[i for i,l in enumerate(lines) if re.match(r'^_Z\w*:',l) and pat in l][0]
end=[i for i,l in enumerate(lines) if i>start and l.startswith('.Lfunc_end')][0]
cur=None; cnt={}; tot={}
for l in lines[start:end]:
    m=re.match(r'^(\.LBB\d+_\d+):(.*)',l)
    if m:
        rest=m.group(2)
        mm=re.search(r'Header=(BB\d+_\d+)',rest)
        cur=mm.group(1) if mm else ('self:'+m.group(1) if 'Loop Header' in rest else None)
        continue
    mm=re.match(r'^; %bb\.\d+:.*Header=(BB\d+_\d+)',l)
    if mm: cur=mm.group(1); continue
    if re.match(r'^; %bb\.\d+:',l): cur=None; continue
    if l.startswith('\t') and not l.strip().startswith(('.',';')):
        k=cur or 'none'
        tot[k]=tot.get(k,0)+1
        if 'scratch_' in l: cnt[k]=cnt.get(k,0)+1
for k in tot: print(k,'instr',tot[k],'scratch',cnt.get(k,0))
