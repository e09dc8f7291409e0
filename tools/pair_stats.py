# Counts of (pixel, face) pairs on the bench workload (numpy model of the rect / inside / conservative-far tests): candidates, walked
# pairs, and the wave iterations of the pair-walk kernel -- unbalanced, rank-paired (the kernel), synchronised over a tile, dense.
#   python tools/pair_stats.py
import numpy as np, sys, math
sys.path.insert(0,'/root/repo')
from lasr_amd import synth
import bench
NU=bench.NU; IS=256
v,f,tex=synth.blobby_mesh(NU)
n_cycle=bench.N_FRAMES_CYCLE
frames=[0,37,101,200]
pv=synth.frame_vertices(v,n_cycle,first=0,count=256)
m=synth.LASR_MODES
thr=math.log(1./m['dist_eps']-1.)*m['sigma_val']
r=math.sqrt(thr)
print('F',f.shape[0],'thr',thr,'r px',r*IS/2)
xs=(2*np.arange(IS)+1-IS)/IS
tot=dict(cand=0,inside=0,near=0,real=0)
stats=[]
ALT=[]
def tri_dist2(px,py,a,b,c):
    def seg(p0,p1):
        d=p1-p0; t=((px-p0[0])*d[0]+(py-p0[1])*d[1])/max(d@d,1e-30); t=np.clip(t,0,1)
        qx=p0[0]+t*d[0]-px; qy=p0[1]+t*d[1]-py; return qx*qx+qy*qy
    return np.minimum(np.minimum(seg(a,b),seg(b,c)),seg(c,a))
for fr in frames:
    fv=pv[fr][f]  # F,3,3
    F=fv.shape[0]
    # per pixel pair lists: arrays of (face, kind) ; kind 0 inside 1 near-outside
    cnt_in=np.zeros((IS,IS),np.int32); cnt_out=np.zeros((IS,IS),np.int32); cnt_cand=np.zeros((IS,IS),np.int32); cnt_real=np.zeros((IS,IS),np.int32)
    per_face=[]
    for k in range(F):
        t=fv[k]; x=t[:,0]; y=t[:,1]
        xmin,xmax,ymin,ymax=x.min()-r,x.max()+r,y.min()-r,y.max()+r
        ix=np.nonzero((xs>=xmin)&(xs<=xmax))[0]; iy=np.nonzero((xs>=ymin)&(xs<=ymax))[0]
        if len(ix)==0 or len(iy)==0: per_face.append(None); continue
        X,Y=np.meshgrid(xs[ix],xs[iy])
        a,b,c=t[0,:2],t[1,:2],t[2,:2]
        det=(b[0]-a[0])*(c[1]-a[1])-(b[1]-a[1])*(c[0]-a[0])
        if abs(det)<1e-12: per_face.append(None); continue
        w0=((b[0]-X)*(c[1]-Y)-(b[1]-Y)*(c[0]-X))/det
        w1=((c[0]-X)*(a[1]-Y)-(c[1]-Y)*(a[0]-X))/det
        w2=1-w0-w1
        inside=(w0>0)&(w1>0)&(w2>0)
        # heights
        def h(p,q,s): # height of p over edge q-s
            e=s-q; return abs(det)/np.linalg.norm(e)
        h0,h1,h2=h(a,b,c),h(b,c,a),h(c,a,b)
        far=(w0*h0<-r*1.0247)|(w1*h1<-r*1.0247)|(w2*h2<-r*1.0247)
        d2=tri_dist2(X,Y,a,b,c)
        real=inside|(d2<thr)
        near=(~inside)&(~far)
        rows=IS-1-iy  # row from top
        per_face.append((ix,rows,inside,near))
        sl=np.ix_(rows,ix)
        cnt_cand[sl]+=1; cnt_in[sl]+=inside; cnt_out[sl]+=near; cnt_real[sl]+=real
    tot['cand']+=cnt_cand.sum(); tot['inside']+=cnt_in.sum(); tot['near']+=cnt_out.sum(); tot['real']+=cnt_real.sum()
    # per tile / chunk stats
    T=16
    for ty in range(IS//T):
        for tx in range(IS//T):
            ent=[]
            for k,pf in enumerate(per_face):
                if pf is None: continue
                ix,rows,inside,near=pf
                if ix[-1]<tx*T or ix[0]>=tx*T+T or rows.min()>=ty*T+T or rows.max()<ty*T: continue
                ent.append(k)
            for c0 in range(0,len(ent),64):
                ch=ent[c0:c0+64]
                ci=np.zeros((T,T),np.int32); co=np.zeros((T,T),np.int32); cc=np.zeros((T,T),np.int32)
                for k in ch:
                    ix,rows,inside,near=per_face[k]
                    mx=(ix>=tx*T)&(ix<tx*T+T); my=(rows>=ty*T)&(rows<ty*T+T)
                    sub=np.ix_(rows[my]-ty*T,ix[mx]-tx*T)
                    ci[sub]+=inside[np.ix_(my,mx)]; co[sub]+=near[np.ix_(my,mx)]; cc[sub]+=1
                # waves: rows w, w+4..
                row=[]
                for w in range(4):
                    kk=(ci+co)[w::4].reshape(-1); cand=cc[w::4].reshape(-1)
                    srt=np.sort(kk)[::-1]
                    bal=np.maximum.reduce([np.ceil((srt[i]+srt[63-i])/2) for i in range(32)])
                    # alternatives: a heavy lane shared by TWO light ones (21 triples + 1 leftover), quads (rank r, 31-r, 32+r, 63-r)
                    tri=max(max(np.ceil((srt[i]+srt[63-2*i]+srt[62-2*i])/3) for i in range(21)), srt[21])
                    quad=max(np.ceil((srt[i]+srt[31-i]+srt[32+i]+srt[63-i])/4) for i in range(16))
                    # the quad scheme with what the kernel can actually move: only OUTSIDE pairs change hands; flows A -> D, B -> C, A -> B
                    ko=co[w::4].reshape(-1); order=np.argsort(-kk, kind='stable'); ks=kk[order]; os_=ko[order]
                    cur_pairs=0; cur_quads=0
                    for i in range(32):
                        a,d=ks[i],ks[63-i]; mv=min((a-d)//2, os_[i]); cur_pairs=max(cur_pairs, a-mv, d+mv)
                    for i in range(16):
                        kA,kB,kC,kD=ks[i],ks[31-i],ks[32+i],ks[63-i]; pA,pB=os_[i],os_[31-i]
                        t=-(-(kA+kB+kC+kD)//4)
                        dD=min(max(0,t-kD),pA); y=min(max(0,t-kC),pB); x=min(max(0,t-kB+y),pA-dD)
                        cur_quads=max(cur_quads, kA-dD-x, kD+dD, kB+x-y, kC+y)
                    ALT.append((tri, quad, np.ceil(kk.sum()/64), cur_pairs, cur_quads))
                    row.append((kk.sum(),kk.max(),bal,cand.max(),cand.sum(),ci[w::4].max(),ci[w::4].sum()))
                stats.append((len(ch),row))
print(tot)
st=stats
nchunks=len(st)
pairs=sum(r[0] for _,rows in st for r in rows)
M=sum(r[1] for _,rows in st for r in rows)
Mb=sum(r[2] for _,rows in st for r in rows)
Mt=sum(4*max(r[2] for r in rows) for _,rows in st)
Cm=sum(r[3] for _,rows in st for r in rows)
Cs=sum(r[4] for _,rows in st for r in rows)
dense=sum(math.ceil(sum(r[0] for r in rows)/256)*4 for _,rows in st)
densew=sum(math.ceil(r[0]/64) for _,rows in st for r in rows)
print('frames',len(frames),'chunks',nchunks,'entries',sum(n for n,_ in st))
print('pairs',pairs,'wave-iters: unbalanced',M,'balanced',Mb,'tile-synced',Mt,'dense per wave',densew,'dense per tile',dense)
print('walk iterations with triples', sum(a[0] for a in ALT), 'quads', sum(a[1] for a in ALT), 'dense', sum(a[2] for a in ALT), '| outside-only: pairs', sum(a[3] for a in ALT), 'quads', sum(a[4] for a in ALT))
print('classify iters (max cand per wave)',Cm,'cand sum/64',Cs/64)
print('inside-mode iterations (max inside pairs per lane of a chunk-wave)',sum(r[5] for _,rows in st for r in rows),'inside pairs / 64',sum(r[6] for _,rows in st for r in rows)/64)
import collections
Ts=np.array([r[0] for _,rows in st for r in rows])
print('chunk-waves',len(Ts),'mean T',Ts.mean(),'pct',np.percentile(Ts,[50,75,90,95,99,100]))
for q in (256,384,512,768,1024): print('T>',q,(Ts>q).mean(), 'pairs share',Ts[Ts>q].sum()/Ts.sum())
nz=Ts[Ts>0]; print('nonzero',len(nz),'batches dense',np.ceil(nz/64).sum(),'ideal',nz.sum()/64)
