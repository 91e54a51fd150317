import sys; sys.path.insert(0,'.')
import torch
from e2enet_medical_amd._lib import lib
L=lib()
K, VX, VY, VZ = 3, 9, 10, 11
agg = torch.zeros((K, VX, VY, VZ), device="cuda"); cnt = torch.zeros_like(agg)
patch = torch.rand((K, 5, 6, 7), generator=torch.Generator().manual_seed(5))
gs = torch.rand((5, 6, 7), generator=torch.Generator().manual_seed(6)) + 0.1
ra, rc = torch.zeros((K, VX, VY, VZ)), torch.zeros((K, VX, VY, VZ))
pd, gd = patch.cuda(), gs.cuda()
for (x0, y0, z0) in [(0, 0, 0), (4, 4, 4), (2, 1, 3)]:
    L.sw_accumulate(pd.data_ptr(), gd.data_ptr(), agg.data_ptr(), cnt.data_ptr(), K, VX, VY, VZ, 5, 6, 7, x0, y0, z0, 0)
    ra[:, x0:x0 + 5, y0:y0 + 6, z0:z0 + 7] += patch * gs
    rc[:, x0:x0 + 5, y0:y0 + 6, z0:z0 + 7] += gs
    torch.cuda.synchronize()
    d=(agg.cpu()-ra).abs(); dc=(cnt.cpu()-rc).abs()
    print((x0,y0,z0), d.max().item(), (d>0).sum().item(), dc.max().item(), (dc>0).sum().item())
