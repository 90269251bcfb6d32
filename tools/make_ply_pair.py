"""Write a synthetic scan pair as PLY files (for tools/register_ply.py): python tools/make_ply_pair.py N out_dir"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lidar-global-registration_amd"))
import numpy as np
from lgr_amd import formats, synthetic
n, out = int(sys.argv[1]), sys.argv[2]
os.makedirs(out, exist_ok=True)
pair = synthetic.make_pair(n, seed=7)
formats.write_ply(os.path.join(out, "src.ply"), pair["src"], with_normals=False)
formats.write_ply(os.path.join(out, "tgt.ply"), pair["tgt"], with_normals=False)
np.savetxt(os.path.join(out, "T_gt.txt"), pair["T_gt"])
print("wrote", out)
