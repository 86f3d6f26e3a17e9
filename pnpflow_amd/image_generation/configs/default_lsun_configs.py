"""The fields of configs/default_lsun_configs.py the restoration path reads (no ml_collections offline: a plain attribute dict)."""


class ConfigDict(dict):
    """dict with attribute access (what the reference uses ml_collections.ConfigDict for)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def get_default_configs():
    config = ConfigDict()
    config.training = ConfigDict(batch_size=64, continuous=True, reduce_mean=False)
    config.sampling = ConfigDict(init_noise_scale=1.0, use_ode_sampler='ode', ode_tol=1e-5, sample_N=1000, sigma_variance=0.0)
    config.data = ConfigDict(dataset='LSUN', image_size=256, random_flip=True, uniform_dequantization=False, centered=False,
                             num_channels=3)
    config.model = ConfigDict(sigma_max=378, sigma_min=0.01, num_scales=2000, beta_min=0.1, beta_max=20., dropout=0.,
                              embedding_type='fourier')
    config.seed = 42
    return config
