import sys, time, cProfile, pstats
sys.path.insert(0, '.')
from peakachu_amd import cool, utils
c = cool.CoolFile('/tmp/e2e.cool')
t=time.time(); X=c.matrix(balance=False, sparse=True).fetch('chr1'); print('fetch raw chr1 %.3f'%(time.time()-t))
c2 = cool.CoolFile('/tmp/e2e.cool')
pr=cProfile.Profile(); pr.enable()
X=c2.matrix(balance='weight', sparse=True).fetch('chr1')
R=c2.matrix(balance=False, sparse=True).fetch('chr1')
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(18)
t=time.time(); A=utils.tocsr(X); B=utils.tocsr(R); print('tocsr x2 %.3f'%(time.time()-t))
