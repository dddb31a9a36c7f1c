"""Option tree for the renderer hot path.

Mirrors the keys the reference renderer reads, with the
defaults of /root/reference/train_settings/default_config.py:88-127 (LLFF-type)
and :247-272 (360-type data: metric depth, 1024 rays).  Reference
`train_settings/*` configs can be passed to `Graph` unchanged; this module only
exists so tests/bench/smoke can build an `opt` without the reference tree.
"""
from .edict import EasyDict as edict


def default_opt(**over):
    o = edict()
    o.max_iter = 200000
    o.mask_img = False
    o.barf_c2f = None
    o.arch = edict(
        layers_feat=[None, 256, 256, 256, 256, 256, 256, 256, 256],
        layers_feat_fine=None,
        layers_rgb=[None, 128, 3],
        skip=[4],
        density_activ="softplus",
        tf_init=True,
        posenc=edict(include_pi_in_posenc=True, add_raw_3D_points=True, add_raw_rays=True,
                     log_sampling=True, L_3D=10, L_view=4),
    )
    o.nerf = edict(
        view_dep=True,
        depth=edict(param="metric", range=[1, 0]),
        sample_intvs=128,
        sample_stratified=True,
        fine_sampling=False,
        sample_intvs_fine=128,
        rand_rays=1024,
        density_noise_reg=False,
        setbg_opaque=False,
    )
    o.camera = edict(model="perspective", ndc=False)
    _merge(o, over)
    return o


def _merge(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = v


def baseline_opt(config=1, **over):
    """BASELINE.json configs: 0 = 256 rays / 64 coarse (+128 fine) CPU case,
    1 = 4096 rays x (64+128), both `nerf_training_w_gt_poses/dtu/nerf.py`
    (fine_sampling, density_noise_reg=True, metric depth, no c2f);
    2 = joint pose/BARF c2f [0.4, 0.7] (`joint_pose_nerf_training/dtu/barf.py`)."""
    base = dict(nerf=dict(fine_sampling=True, sample_intvs=64, sample_intvs_fine=128,
                          density_noise_reg=True, depth=dict(param="metric")))
    if config == 0:
        base["nerf"]["rand_rays"] = 256
    elif config == 1:
        base["nerf"]["rand_rays"] = 4096
    elif config == 2:
        base["nerf"]["rand_rays"] = 4096
        base["nerf"]["density_noise_reg"] = False
        base["barf_c2f"] = [0.4, 0.7]
    else:
        raise ValueError(config)
    o = default_opt(**base)
    _merge(o, over)
    return o
