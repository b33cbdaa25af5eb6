#!/usr/bin/env python3
"""Node visits of the tree walk on the C3 mesh (oracle tree, CPU): primary and Lambertian-scattered rays, with and without a
normal-cone test per subtree (skip a subtree whose stored normals all have d.n < 0).  python tools/cone_sim.py"""
import sys, time, tempfile
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cudaraytracer_amd import scenes
from oracle import pyoracle as po
path,nx,ny,spp=scenes.baseline_obj("C3", tempfile.mkdtemp())
pts,idx=po.parse_obj(path); w=po.World(pts,idx,None,1)
st=po.fresh_states(8,8); bvh=po.Bvh(w,st)
nodes=bvh.nodes(); order=bvh.order()
print(nodes.dtype, len(nodes))
box=nodes['box'].reshape(-1,6).astype(np.float64); skip=nodes['skip']; first=nodes['first']; count=nodes['count']; leaf=nodes['isLeaf']
# triangles in reversed winding: v0=p[i2], v1=p[i1], v2=p[i0]; normal = unit(cross(v1-v0, v2-v0))
t=(idx-1).astype(int); P=pts.astype(np.float64)
v0,v1,v2=P[t[:,2]],P[t[:,1]],P[t[:,0]]
n=np.cross(v1-v0,v2-v0); n/= np.linalg.norm(n,axis=1)[:,None]
ns=n[order]  # slot order
# per-node cone: axis & min cos over its slots (nodes store first slot,count only for leaves; compute ranges bottom-up via preorder: subtree slots = [first_slot(i), first_slot(skip(i)))
N=len(nodes)
# first slot of subtree: for leaf = first; for inner = first slot of left child (i+1)
fs=np.zeros(N+1,dtype=int); 
for i in range(N-1,-1,-1):
    fs[i]= first[i] if leaf[i] else fs[i+1]
fs[N]=len(order)
def subtree_range(i):
    j=skip[i]; return fs[i], (fs[j] if j<N else len(order))
axis=np.zeros((N,3)); mincos=np.zeros(N)
for i in range(N):
    a,b=subtree_range(i); m=ns[a:b]; ax=m.sum(0); l=np.linalg.norm(ax)
    if l<1e-12: mincos[i]=-1; continue
    ax/=l; axis[i]=ax; mincos[i]=(m@ax).min()
sinth=np.sqrt(np.maximum(0,1-np.minimum(mincos,1)**2))
def box_hit(i,o,d):
    tmin,tmax=0.001,3.4e38
    for a in range(3):
        with np.errstate(divide='ignore',invalid='ignore'):
            qa=(box[i,a]-o[a])/d[a]; qb=(box[i,3+a]-o[a])/d[a]
        t0=min(qa,qb); t1=max(qa,qb)
        tmin=max(t0,tmin); tmax=min(t1,tmax)
        if tmax<=tmin: return False
    return True
def traverse(o,d,cone):
    i=0; visited=0; leaves=0
    while i<N:
        visited+=1
        hit=box_hit(i,o,d)
        if hit and cone and mincos[i]>0:
            # all normals within theta of axis: d.n_i<0 for all iff angle(d,axis)>90+theta  <=> d.axis < -sin(theta)
            if d@axis[i] < -sinth[i]-1e-4: hit=False
        if hit and leaf[i]: leaves+=1
        i = i+1 if (hit and not leaf[i]) else skip[i]
    return visited,leaves
cam=po.make_camera((0,100,1000),(0,150,0),(0,1,0),40.0,np.float32(nx)/np.float32(ny),0.0,10.0)
llc=np.array(cam.lower_left_corner[:]); H=np.array(cam.horizontal[:]); V=np.array(cam.vertical[:]); O=np.array(cam.origin[:])
rng=np.random.RandomState(1)
res={'prim':[], 'prim_cone':[], 'sec':[], 'sec_cone':[]}
e1=v1-v0; e2=v2-v0
k=0
while k<60:
    u,v=rng.uniform(0.35,0.65),rng.uniform(0.1,0.55)
    d=llc+u*H+v*V-O; d/=np.linalg.norm(d)
    # brute force hit
    h=np.cross(d,e2); a=(e1*h).sum(1); ok=np.abs(a)>1e-6; f=1/np.where(ok,a,1); s=O-v0; uu=f*(s*h).sum(1); q=np.cross(s,e1); vv=f*(q@d); tt=f*(e2*q).sum(1)
    m=ok&(uu>=0)&(uu<=1)&(vv>=0)&(uu+vv<=1)&((n@d)>=0)
    if not m.any(): continue
    j=np.argmin(np.where(m,tt,1e30)); p=O+tt[j]*d
    res['prim'].append(traverse(O,d,False)[0]); res['prim_cone'].append(traverse(O,d,True)[0])
    # lambertian scatter: target = p + normal + rnd ; normal = stored normal (points inward?) use shading normal n[j]
    r=rng.normal(size=3); r/=np.linalg.norm(r); r*=rng.uniform()**(1/3)
    sd=n[j]+r; sd/=np.linalg.norm(sd)
    res['sec'].append(traverse(p,sd,False)[0]); res['sec_cone'].append(traverse(p,sd,True)[0])
    k+=1
for kk,vv_ in res.items(): print(kk, np.mean(vv_), np.median(vv_), np.max(vv_))
