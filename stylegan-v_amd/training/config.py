"""Model hyper-parameter sets for the hot-path benchmarks and tests.

The reference composes these with hydra/omegaconf (configs/model/stylegan-v.yaml,
configs/sampling/*.yaml) and derives per-resolution values in src/train.py:138-200.  Neither
package exists here and the launch machinery is out of scope, so the values are stated directly.
``Config`` is a plain attribute-access dict that serves the model code the way an omegaconf
``DictConfig`` does (``cfg.motion.v_dim``, ``cfg.get('x', d)``, ``**cfg.sampling``).
"""

import copy


class Config(dict):
    """dict with attribute access; nested dicts are converted on construction."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    def __setitem__(self, key, value):
        if isinstance(value, dict) and not isinstance(value, Config):
            value = Config(value)
        super().__setitem__(key, value)

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError as exc:
            raise AttributeError(name) from exc

    __setattr__ = __setitem__

    def __deepcopy__(self, memo):
        return Config({k: copy.deepcopy(v, memo) for k, v in self.items()})

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, Config) else v) for k, v in self.items()}


def sampling_config(num_frames_per_video=3, max_num_frames=1024, max_dist=32):
    """configs/sampling/random.yaml + base.yaml (random frame sampling, 3 frames per video)."""
    return Config(type='random', num_frames_per_video=num_frames_per_video, max_num_frames=max_num_frames,
                  total_dists=[1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048], max_dist=max_dist)


def generator_config(sampling=None, w_dim=512, z_dim=512, motion_dim=512, time_enc_dim=256, min_period_len=16,
                     max_period_len=1024, kernel_size=11, c_dim=0):
    """configs/model/stylegan-v.yaml:4-45 (motion_z_distance == time_enc.min_period_len, :17)."""
    return Config(
        sampling=sampling or sampling_config(), use_noise=False, input=dict(type='temporal'), w_dim=w_dim, z_dim=z_dim, c_dim=c_dim,
        motion=dict(z_dim=motion_dim, v_dim=motion_dim, motion_z_distance=min_period_len, gen_strategy='conv', kernel_size=kernel_size,
                    use_fractional_t=True, fourier=True),
        time_enc=dict(cond_type='concat_const', dim=time_enc_dim, min_period_len=min_period_len, max_period_len=max_period_len,
                      phase_dropout_std=1.0))


def discriminator_config(sampling=None):
    """configs/model/stylegan-v.yaml:47-51."""
    return Config(sampling=sampling or sampling_config(), concat_res=16, num_frames_div_factor=2, dummy_c=False)


def model_kwargs(resolution=256, batch_size=32, num_gpus=1, fp32=True, min_period_len=16, num_frames_per_video=3, lowp_dtype=None):
    """Constructor arguments of G and D as src/train.py:138-200 derives them for cfg='auto'.

    fmaps 0.5 below 512^2 else 1 (:158), mapping depth 2 (:139), mbstd group min(batch_gpu, 4) (:157),
    r1_gamma 0.0002*res^2/batch (:160), lr 0.0025 (0.002 at 1024^2, :159), ema_kimg batch*10/32 (:161).
    """
    fmaps = 1.0 if resolution >= 512 else 0.5
    samp = sampling_config(num_frames_per_video=num_frames_per_video)
    gcfg = generator_config(samp, min_period_len=min_period_len)
    dcfg = discriminator_config(samp)
    synth = dict(channel_base=int(fmaps * 32768), channel_max=512, num_fp16_res=0 if fp32 else 4, conv_clamp=None if fp32 else 256)
    block_kwargs = {}
    if not fp32 and lowp_dtype is not None:  # reference mixed precision is fp16 (networks.py:227,461); bf16 is this build's extension
        synth['lowp_dtype'] = lowp_dtype
        block_kwargs['lowp_dtype'] = lowp_dtype
    g_kwargs = dict(c_dim=0, w_dim=512, img_resolution=resolution, img_channels=3, mapping_kwargs=dict(num_layers=2, cfg=gcfg),
                    synthesis_kwargs=synth, cfg=gcfg)
    d_kwargs = dict(c_dim=0, img_resolution=resolution, img_channels=3, channel_base=int(fmaps * 32768), channel_max=512,
                    num_fp16_res=0 if fp32 else 4, conv_clamp=None if fp32 else 256, block_kwargs=block_kwargs, mapping_kwargs=dict(num_layers=2),
                    epilogue_kwargs=dict(mbstd_group_size=min(batch_size // num_gpus, 4)), cfg=dcfg)
    train = dict(r1_gamma=0.0002 * resolution ** 2 / batch_size, lr=0.002 if resolution >= 1024 else 0.0025, betas=(0.0, 0.99),
                 ema_kimg=batch_size * 10 / 32, ema_rampup=0.05, G_reg_interval=4, D_reg_interval=16, pl_weight=0.0)
    return g_kwargs, d_kwargs, Config(train)


def small_test_configs():
    """The miniature G/D hyper-parameters the golden fixtures were generated with (tests/golden/make_golden.py)."""
    samp = Config(type='random', num_frames_per_video=3, max_num_frames=64, total_dists=[1, 2, 4, 8, 16, 32], max_dist=32)
    gcfg = Config(sampling=samp, use_noise=False, input=dict(type='temporal'), w_dim=32, z_dim=32, c_dim=0,
                  motion=dict(z_dim=24, v_dim=24, motion_z_distance=4, gen_strategy='conv', kernel_size=5, use_fractional_t=True, fourier=True),
                  time_enc=dict(cond_type='concat_const', dim=8, min_period_len=4, max_period_len=64, phase_dropout_std=1.0))
    dcfg = Config(sampling=samp, concat_res=16, num_frames_div_factor=2, dummy_c=False)
    return gcfg, dcfg


def small_test_model_kwargs(res=32):
    gcfg, dcfg = small_test_configs()
    g_kwargs = dict(c_dim=0, w_dim=32, img_resolution=res, img_channels=3, mapping_kwargs=dict(num_layers=2, cfg=gcfg),
                    synthesis_kwargs=dict(channel_base=res * 16, channel_max=32, num_fp16_res=0, conv_clamp=None), cfg=gcfg)
    d_kwargs = dict(c_dim=0, img_resolution=res, img_channels=3, channel_base=res * 16, channel_max=32, num_fp16_res=0, conv_clamp=None,
                    mapping_kwargs=dict(num_layers=2), epilogue_kwargs=dict(mbstd_group_size=2), cfg=dcfg)
    return g_kwargs, d_kwargs


def mid_test_configs():
    """Hyper-parameters of the mid-size golden (tests/golden/networks_mid.npz): every conv has 128 output channels and
    the network reaches 128^2, so that G / D / R1 parity on the GPU runs THROUGH the MFMA convolution kernels, the
    whole-tile GEMM and the wide (>= 129 column) upfirdn2d kernels instead of their small-shape fallbacks."""
    samp = Config(type='random', num_frames_per_video=3, max_num_frames=64, total_dists=[1, 2, 4, 8, 16, 32], max_dist=32)
    gcfg = Config(sampling=samp, use_noise=False, input=dict(type='temporal'), w_dim=64, z_dim=64, c_dim=0,
                  motion=dict(z_dim=24, v_dim=24, motion_z_distance=4, gen_strategy='conv', kernel_size=5, use_fractional_t=True, fourier=True),
                  time_enc=dict(cond_type='concat_const', dim=8, min_period_len=4, max_period_len=64, phase_dropout_std=1.0))
    dcfg = Config(sampling=samp, concat_res=16, num_frames_div_factor=2, dummy_c=False)
    return gcfg, dcfg


def mid_test_model_kwargs(res=128, channels=128):
    gcfg, dcfg = mid_test_configs()
    g_kwargs = dict(c_dim=0, w_dim=64, img_resolution=res, img_channels=3, mapping_kwargs=dict(num_layers=2, cfg=gcfg),
                    synthesis_kwargs=dict(channel_base=res * channels, channel_max=channels, num_fp16_res=0, conv_clamp=None), cfg=gcfg)
    d_kwargs = dict(c_dim=0, img_resolution=res, img_channels=3, channel_base=res * channels, channel_max=channels, num_fp16_res=0, conv_clamp=None,
                    mapping_kwargs=dict(num_layers=2), epilogue_kwargs=dict(mbstd_group_size=2), cfg=dcfg)
    return g_kwargs, d_kwargs
