import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch, oracle_lib
from poolgen_amd import Engine, synth
o=oracle_lib.load(); eng=Engine(0)
for p,n in [(5000,200),(4099,100),(130,5)]:
    G=synth.genotype_matrix(p,n,'cuda',seed=3); Y=synth.phenotypes(G,n,k=2,seed=3)
    eng.covariates_set(n,None,Y); b,v,pv=(x.cpu().numpy() for x in eng.ols_sweep(G,2,n))
    ref=o.ols_with_covariate(G.cpu().numpy(),Y,force_m=0,n=n)
    g=G.cpu().numpy()[:,:n].astype(np.longdouble); y=Y.astype(np.longdouble)
    gc=g-g.mean(1,keepdims=True); yc=y-y.mean(0,keepdims=True)
    sgg=(gc*gc).sum(1); bt=(gc@yc)/sgg[:,None]
    rss=(yc*yc).sum(0)[None,:]-bt*(gc@yc); vt=rss/(n-2)/sgg[:,None]
    rel=lambda a,b: float(np.max(np.abs(a-b)/np.abs(b)))
    print(p,n,"beta gpu-vs-ld %.2e oracle-vs-ld %.2e gpu-vs-oracle %.2e"%(rel(b,bt),rel(ref['beta'],bt),rel(b,ref['beta'])))
    print(p,n,"var  gpu-vs-ld %.2e oracle-vs-ld %.2e gpu-vs-oracle %.2e"%(rel(v,vt),rel(ref['var'],vt),rel(v,ref['var'])))
    print(p,n,"pval abs gpu-vs-oracle %.2e"%np.max(np.abs(pv-ref['pval'])))
    i=np.argmax(np.abs(b-bt)/np.abs(bt)); print("worst beta idx",i,b.ravel()[i],ref['beta'].ravel()[i],float(bt.ravel()[i]))
