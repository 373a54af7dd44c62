import sys, torch, time
sys.path.insert(0,'.')
from geoa3_amd import pointnet2 as p2
for (B,N,m) in [(250,1024,512),(250,512,128),(1,1024,512),(250,1024,64)]:
    x=torch.randn(B,N,3,device='cuda')
    for _ in range(2): p2.ext.furthest_point_sampling(x,m)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(5): p2.ext.furthest_point_sampling(x,m)
    torch.cuda.synchronize(); print(B,N,m,(time.perf_counter()-t)/5*1e6,'us')
