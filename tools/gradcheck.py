import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch, parity
from hip_adapter import editnet_modules, to_dev
from show_edit_tell_amd.train import xe_loss_sum
d, xe, rl = editnet_modules("editnet_small"); g = parity.load("editnet_small"); xe.eval()
pred, caps_s, dl, _ = xe(to_dev(d["X"]), to_dev(d["caps"]), to_dev(d["clen"]), to_dev(d["prev"]), to_dev(d["plen"]), False, 0.0)
ls, n, _, _ = xe_loss_sum(pred, caps_s, dl); (ls/n).backward()
for k,p in xe.named_parameters():
    got=p.grad.cpu().numpy(); ref=g["grad."+k]; print("%-55s max|ref| %.3e  err %.3e  rel %.2e"%(k, np.abs(ref).max(), np.abs(got-ref).max(), np.abs(got-ref).max()/max(np.abs(ref).max(),1e-12)))
