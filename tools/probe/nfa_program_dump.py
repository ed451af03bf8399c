import sys
import os; R=os.path.join(os.path.dirname(os.path.abspath(__file__)),'..','..'); sys.path[:0]=[R, os.path.join(R,'tests')]
import numpy as np, roaringregex_amd as rr
from patterns import U2
def dump(p):
    r=rr.RRegex(p, rr.ENGINE_NFA); w=[int(x) for x in r.program(rr.ENGINE_NFA)]
    W,nbits,nexc=w[0],w[1],w[2]; o=4
    def big(ws):
        v=0
        for i,x in enumerate(ws): v|=x<<(32*i)
        return v
    init,fin,chain,self_,excm,cgrp,ctgt=[big(w[o+i*W:o+(i+1)*W]) for i in range(7)]; o+=7*W
    B=[big(w[o+c*W:o+(c+1)*W]) for c in range(256)]; o+=256*W
    X=[big(w[o+b*W:o+(b+1)*W]) for b in range(nbits)]
    bits=lambda v:[i for i in range(nbits) if (v>>i)&1]
    print(p[:50],"W",W,"nbits",nbits,"nexc",nexc)
    print(" self",bits(self_)); print(" cgrp",bits(cgrp)); print(" ctgt",bits(ctgt)); print(" fin",bits(fin))
    dist={}
    for b in bits(excm):
        t=bits(X[b]); print("  exc",b,"->",t)
        for q in t: dist.setdefault(q-b,[]).append((b,q))
    print(" distances:",{d:len(v) for d,v in sorted(dist.items())})
    # distinct target sets
    ts={}
    for b in bits(excm): ts.setdefault(X[b],[]).append(b)
    print(" distinct target sets:",len(ts), [(v,bits(k)) for k,v in ts.items()])
dump(U2)
dump(U2+"(x|y)*x(x|y){30}")
