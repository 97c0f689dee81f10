import sys
src=open('/root/repo/tools/kernel_probe.py').read().split('if __name__')[0]
exec(compile(src,'kp','exec'))
for (M,N,K) in [(35100,512,512),(35100,1536,512),(17550,2048,2048),(17550,3072,3072),(35100,3072,2048),(35100,3072,3072),(17550,2048,3072)]:
    gemm_case(M,N,K)
for t in (0,1):
    import os
    os.environ['BYA_GEMM_TILE']=str(t)
