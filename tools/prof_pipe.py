import sys, ctypes as C
sys.path.insert(0,'/root/repo')
import numpy as np, torch
from cool_chic_amd import DecodeBatch
from cool_chic_amd._lib import lib
from oracle import oracle_py as O
bs=open('/root/repo/tests/golden/kodim14.cool','rb').read()
fh,ccs=O.split_stream(bs)[1][0]
b=DecodeBatch(0)
n=int(sys.argv[1]) if len(sys.argv)>1 else 1
for i in range(n): b.add(*ccs[0],8,0)
import time
for rep in range(2):
    torch.cuda.synchronize(); t=time.time(); b.run(stage=0); b.wait(); dt=time.time()-t
print('entropy stage wall ms', dt*1e3)
st=np.zeros(64,np.int32); lib().ccd_batch_slot_stats(b._h,0,st.ctypes.data)
u=st[4:24].view(np.uint64)
print('decoder : total %d wait %d decode %d'%(u[0],u[1],u[2]), ' per symbol decode cycles %.1f'%(u[2]/526272), 'wait frac %.2f'%(u[1]/u[0]))
print('decoder outside grids: ifce+setup %d, end-of-grid barrier %d'%(u[3],u[4]))
print('producer1: total %d wait %d gather %d mlp %d table %d'%(u[5],u[6],u[7],u[8],u[9]))
nb=526272/16/7
print(' per batch (approx %d batches): gather %.0f mlp %.0f table %.0f cycles'%(nb,u[7]/nb,u[8]/nb,u[9]/nb))

hw=[(512,768),(256,384),(128,192),(64,96)]
print(' IFCE feature pass before grids 0..3 (Mticks, level-3 builds):', [round(int(st[24+2*g])*1024/1e6,3) for g in range(4)])
for g in range(0):
    w,k=int(st[24+2*g])*1024,int(st[25+2*g])*1024
    n=hw[g][0]*hw[g][1]
    print(' grid %d (%dx%d): wait %.1fM work %.1fM cycles -> %.0f cycles/symbol total, wait share %.2f'%(g,hw[g][0],hw[g][1],w/1e6,k/1e6,(w+k)/n,w/(w+k+1)))

print(' grid0 steady-state decoder wait by batch position j (Mcycles):', [round(int(x)*1024/1e6,2) for x in st[32:38]])

e=st[40:50].view(np.uint64)
nt=max(int(e[4]),1)
print(' producer1 tasks %d: per task gather %.0f mlp %.0f (reload %.0f stab %.0f hidden %.0f out+meta %.0f) table %.0f wait %.0f'%(nt,u[7]/nt,u[8]/nt,e[0]/nt,e[1]/nt,e[2]/nt,e[3]/nt,u[9]/nt,u[6]/nt))

print(' light counters (decoder, per grid): total Mticks, stalled Mticks, stall events')
for g in range(4):
    print('  grid %d: total %.1fM stalled %.1fM (%.0f%%) in %d stalls'%(g,int(st[50+3*g])*1024/1e6,int(st[51+3*g])*1024/1e6,100.0*int(st[51+3*g])/max(int(st[50+3*g]),1),int(st[52+3*g])))
