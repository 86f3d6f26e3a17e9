"""configs/rectified_flow/celeba_hq_pytorch_rf_gaussian.py of the reference: the NCSN++ hyper-parameters of the rectified-flow net."""
from ..default_lsun_configs import get_default_configs


def get_config():
    config = get_default_configs()
    config.training.update(sde='rectified_flow', continuous=False, reduce_mean=True)
    config.sampling.update(method='rectified_flow', init_type='gaussian', init_noise_scale=1.0, use_ode_sampler='rk45')
    config.data.update(dataset='CelebA-HQ-Pytorch', centered=True)
    config.model.update(name='ncsnpp', scale_by_sigma=True, ema_rate=0.999, normalization='GroupNorm', nonlinearity='swish', nf=128,
                        ch_mult=(1, 1, 2, 2, 2, 2, 2), num_res_blocks=2, attn_resolutions=(16,), resamp_with_conv=True,
                        conditional=True, fir=True, fir_kernel=[1, 3, 3, 1], skip_rescale=True, resblock_type='biggan',
                        progressive='output_skip', progressive_input='input_skip', progressive_combine='sum',
                        attention_type='ddpm', init_scale=0., fourier_scale=16, conv_size=3)
    return config
