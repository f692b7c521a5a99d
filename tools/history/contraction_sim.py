#!/usr/bin/env python
"""Design study ("chain contraction"): how many rows per sweep would the NR tree kernel need if degree-2
chains were contracted?  Schedules, on the real topologies, (a) the shipped Hu schedule (rows = radius of the center-rooted
forest), (b) general rake-and-compress list scheduling (any node with <= 1 alive child may be eliminated; no two adjacent
nodes in one row), (c) one level of chain halving (every other pure degree-2 node eliminated in a pre-pass).  Host-side
arithmetic only; nothing here runs on the GPU or is imported by the product."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mapdn_amd.netspec import make_case

def tree_of(net):
    nb=net.n_bus; adj=[[] for _ in range(nb)]
    for f,t,s in zip(net.line_from_bus,net.line_to_bus,net.line_in_service):
        if s: adj[f].append(int(t)); adj[t].append(int(f))
    return adj

def center_forest(adj, slack):
    nb=len(adj); par={}
    import collections
    def bfs(src):
        dist={src:0}; frm={src:-1}; q=[src]
        for u in q:
            for w in adj[u]:
                if w!=slack and w not in dist: dist[w]=dist[u]+1; frm[w]=u; q.append(w)
        far=max(q,key=lambda u:(dist[u],-u)); return far,dist,frm,q
    roots=[]
    for s0 in sorted(adj[slack]):
        a,_,_,_=bfs(s0); b,dist,frm,comp=bfs(a)
        c=b
        for i in range(dist[b]//2): c=frm[c]
        # root at c
        par[c]=-1; st=[c]; seen={c}
        while st:
            u=st.pop()
            for w in adj[u]:
                if w!=slack and w not in seen: seen.add(w); par[w]=u; st.append(w)
        roots.append(c)
    return par

def hu_rows(par,W):
    nodes=list(par); ch={k:[] for k in nodes}
    for k,p in par.items():
        if p>=0: ch[p].append(k)
    depth={}
    def d(k):
        if k not in depth: depth[k]=0 if par[k]<0 else d(par[k])+1
        return depth[k]
    for k in nodes: d(k)
    pend={k:len(ch[k]) for k in nodes}; ready=[k for k in nodes if pend[k]==0]; done=0; R=0
    while done<len(nodes):
        ready.sort(key=lambda k:-depth[k]); take=ready[:W]; ready=ready[W:]
        for k in take:
            done+=1; p=par[k]
            if p>=0:
                pend[p]-=1
                if pend[p]==0: ready.append(p)
        R+=1
    return R, max(depth.values())+1

def rc_rows(par,W,verbose=False):
    # rake+compress greedy list scheduling
    nodes=list(par); par=dict(par)
    ch={k:set() for k in nodes}
    for k,p in par.items():
        if p>=0: ch[p].add(k)
    alive=set(nodes); R=0; ncomp=0
    def height(k,memo={}):
        return 0
    while alive:
        # subtree heights in current contracted tree
        h={}
        order=[]
        st=[k for k in alive if par[k]<0]
        while st:
            u=st.pop(); order.append(u); st.extend(ch[u])
        for u in reversed(order): h[u]=1+max([h[c] for c in ch[u]],default=0)
        dep={}
        for u in order: dep[u]=0 if par[u]<0 else dep[par[u]]+1
        cand=[k for k in alive if len(ch[k])<=1]
        # priority: nodes on the longest paths first: key = dep+h (path length through node), then compress preferred on long chains
        cand.sort(key=lambda k:(-(dep[k]+h[k]), -dep[k]))
        chosen=[]; blocked=set()
        for k in cand:
            if len(chosen)>=W: break
            if k in blocked: continue
            chosen.append(k); blocked.add(k)
            if par[k]>=0: blocked.add(par[k])
            for c in ch[k]: blocked.add(c)
        for k in chosen:
            p=par[k]; cs=list(ch[k])
            if cs:
                ncomp+=1; c=cs[0]; par[c]=p
                if p>=0: ch[p].discard(k); ch[p].add(c)
            else:
                if p>=0: ch[p].discard(k)
            alive.discard(k)
        R+=1
    return R,ncomp

def halve(par):
    ch={k:[] for k in par}
    for k,p in par.items():
        if p>=0: ch[p].append(k)
    comp=set()
    # process top-down along chains so alternate nodes get picked: order by depth
    depth={}
    def d(k):
        if k not in depth: depth[k]=0 if par[k]<0 else d(par[k])+1
        return depth[k]
    for k in par: d(k)
    for k in sorted(par,key=lambda k:depth[k]):
        if par[k]>=0 and len(ch[k])==1 and par[k] not in comp and ch[k][0] not in comp:
            # also require child not compress (checked) ; pick
            comp.add(k)
    newpar={}
    for k,p in par.items():
        if k in comp: continue
        while p in comp: p=par[p]
        newpar[k]=p
    return newpar, comp

if __name__ == "__main__":
    for name in ("case33", "case141", "case141_deep", "case322"):
        net, _ = make_case(name); adj = tree_of(net); par = center_forest(adj, net.ext_grid_bus)
        halved, comp = halve(par)
        print(name, "n", len(par))
        for W in (16, 32):
            print(f"  {W:2d} workers: Hu rows {hu_rows(par, W)[0]:2d} | rake+compress rows {rc_rows(par, W)[0]:2d} ({rc_rows(par, W)[1]} compress steps)"
                  f" | one halving: {len(comp)} nodes in {-(-len(comp) // W)} pre-pass turns, then {hu_rows(halved, W)[0]} rows over {len(halved)} nodes")
