import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch, numpy as np
import id_diff_amd
from helpers import *
from id_diff_amd import _lib, dim_reduction, sde_lib
from id_diff_amd.models import utils as mutils
from oracle import dim as odim, models as omodels, sde as osde
cfg = ncsnpp_config(**{"model.init_scale": 1.0, "model.attn_resolutions": (8,), "data.image_size": 16, "data.effective_image_size": 16, "data.shape": [3, 16, 16], "model.num_res_blocks": 1})
torch.manual_seed(0)
ref_model = omodels.create_model(cfg); model = mutils.create_model(cfg); model.load_state_dict(ref_model.state_dict()); model.to("cuda")
x = torch.rand(3, 16, 16, generator=torch.Generator().manual_seed(1))
sde_c, sde_h = osde.VESDE(0.01, 50, 1000), sde_lib.VESDE(0.01, 50, 1000)
num_batches, _, rows = odim.batching(tuple(x.shape), 100)
noise = torch.randn(num_batches, 100, *x.shape, generator=torch.Generator().manual_seed(7))
S_ref = odim.score_matrix(osde.get_score_fn(sde_c, ref_model), sde_c, x, 100, 1e-5, noise=noise)
b = dim_reduction.ScoreMatrixBuilder(mutils.get_score_fn(sde_h, model), sde_h, 1e-5, torch.device("cuda"))
S = b.build(x.cuda(), 100, noise=noise.reshape(-1, *x.shape)[:rows].cuda())
print("S rel err", rel_err(S.cpu(), S_ref))
# the network output itself on the same perturbed batch
xb = (x + 0.01 * noise[0]).contiguous(); t = torch.full((100,), 1e-5)
with torch.no_grad():
    o_ref = ref_model.eval()(xb, t * 999); o = model(xb.cuda(), (t * 999).cuda()).cpu()
print("model out rel err", rel_err(o, o_ref), " |out| rms", float(o_ref.pow(2).mean().sqrt()), " max", float(o_ref.abs().max()))
# in fp64 the oracle's own rounding
ref64 = omodels.create_model(cfg).double(); ref64.load_state_dict({k: v.double() for k, v in ref_model.state_dict().items()})
with torch.no_grad():
    o64 = ref64.eval()(xb.double(), (t * 999).double())
print("oracle fp32 vs fp64", rel_err(o_ref, o64), " hip fp32 vs fp64", rel_err(o, o64))
