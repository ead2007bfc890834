import sys, zlib, io, math
sys.path.insert(0,'/root/repo')
import numpy as np
from oracle import png_pack, oracle_py
from PIL import Image
oracle_py.build()
bs=open('/root/repo/tests/golden/kodim14.cool','rb').read()
pl=oracle_py.decode_video(bs)[0]['planes']
img=np.stack(pl,axis=-1).astype(np.uint8)
scan,_=png_pack.filter_rows(img)
flat=scan.reshape(-1)
print('scan bytes',flat.size)
buf=io.BytesIO(); Image.fromarray(img).save(buf,format='PNG'); print('PIL',len(buf.getvalue()))
print('zlib6 on our filtered', len(zlib.compress(flat.tobytes(),6)), 'zlib9', len(zlib.compress(flat.tobytes(),9)))
ours=png_pack.pack_rgb8(np.stack(pl).astype(np.uint8)); print('ours literal-only',len(ours))

LBASE=[3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258]
LEXT=[0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0]
DBASE=[1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577]
DEXT=[0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13]
def lsym(l):
    i=np.searchsorted(LBASE,l,side='right')-1; return 257+i, LEXT[i]
def dsym(d):
    i=np.searchsorted(DBASE,d,side='right')-1; return i, DEXT[i]
def huff_bits(hist):
    h=np.array([c for c in hist if c>0],dtype=float)
    if len(h)<2: return float(h.sum())
    lens=png_pack.code_lengths(hist) if len(hist)==257 else None
    # entropy bound is fine for estimation
    p=h/h.sum(); return float(-(h*np.log2(p)).sum())
def cost_block(data, matches):
    # matches: dict pos->(len,dist) greedy parse
    n=len(data); i=0
    lit=np.zeros(286,int); dist=np.zeros(30,int); extra=0
    while i<n:
        m=matches[i] if matches is not None else (0,0)
        if m[0]>=3:
            s,e=lsym(m[0]); lit[s]+=1; extra+=e
            s,e=dsym(m[1]); dist[s]+=1; extra+=e
            i+=m[0]
        else:
            lit[data[i]]+=1; i+=1
    lit[256]+=1
    return huff_bits(lit)+huff_bits(dist)+extra+ 17+ (286+30)*1.5  # rough header

def best_matches_fixed(data, dists, minlen=3, maxlen=258):
    n=len(data); best_len=np.zeros(n,int); best_d=np.zeros(n,int)
    for d in dists:
        if d>=n: continue
        eq=np.zeros(n,bool); eq[d:]=data[d:]==data[:-d]
        # run length of eq starting at i
        run=np.zeros(n+1,int)
        for i in range(n-1,-1,-1):
            run[i]=run[i+1]+1 if eq[i] else 0
        run=np.minimum(run[:n],maxlen)
        better=run>best_len
        best_len[better]=run[better]; best_d[better]=d
    return [(int(l),int(d)) if l>=minlen else (0,0) for l,d in zip(best_len,best_d)]

def best_matches_hash(data, K=8, minlen=3, maxlen=258, window=32768):
    n=len(data); out=[(0,0)]*n; head={}
    for i in range(n-2):
        key=bytes(data[i:i+3]); cand=head.get(key,[])
        bl,bd=0,0
        for j in reversed(cand[-K:]):
            if i-j>window: break
            l=0
            while l<maxlen and i+l<n and data[j+l]==data[i+l]: l+=1
            if l>bl: bl,bd=l,i-j
        if bl>=minlen: out[i]=(bl,bd)
        head.setdefault(key,[]).append(i)
    return out

w=img.shape[1]; stride=3*w+1; rpb=png_pack.rows_per_block(w)
nblk=(img.shape[0]+rpb-1)//rpb
tot={'lit':0,'fixed':0,'hash8':0,'hash8lazy':0}
import time
for b in range(0,nblk,6):   # sample every 6th block
    data=scan[b*rpb:(b+1)*rpb].reshape(-1)
    tot['lit']+=cost_block(data,None)
    t=time.time()
    fixed=[1,2,3,4,5,6,7,8,9,12,15,stride-3,stride-2,stride-1,stride,stride+1,stride+2,stride+3,2*stride]
    tot['fixed']+=cost_block(data,best_matches_fixed(data,fixed))
    tot['hash8']+=cost_block(data,best_matches_hash(data,8))
    print(b, {k:int(v/8) for k,v in tot.items()}, round(time.time()-t,1), flush=True)
print({k:v/tot['lit'] for k,v in tot.items()})
