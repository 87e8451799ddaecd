"""AMPSeptBuilder -- the PACER policy / critic / discriminator network on the MI355X GEMM kernels.

Mirror of /root/reference/pacer/pacer/learning/amp_network_sept_builder.py:14-139 (AMPSeptBuilder.Network) over
amp_network_builder.py:16-124 (AMPBuilder.Network: sigma parameter, discriminator) and network_builder.py:165-277
(A2CBuilder.Network: actor / critic MLPs, value and mu heads) for the configuration the repo ships
(`amp_humanoid_smpl_sept_task.yaml`: continuous actions, fixed sigma, separate critic, no CNN / RNN / D2RL / people
branch).  `load(params)` takes the yaml's `network` dict, `build(name, **kwargs)` the same kwargs rl_games passes
(`actions_num, input_shape, amp_input_shape, self_obs_size, task_obs_size, task_obs_size_detail, mean_std`).
Module and parameter names match the reference, so its checkpoints' `a2c_network.*` entries load with
`load_state_dict` (state_dict keys: `actor_mlp.{0,2}`, `critic_mlp.{0,2}`, `_task_mlp.{0,2}`, `_disc_mlp.{0,2}`,
`_disc_logits`, `mu`, `value`, `sigma`).

Every Linear(+ReLU) is one `emloco_gemm_f32` launch with the bias / ReLU epilogue (fp32 MFMA) through
`predictor.ops.linear`, which also carries the backward (config 2 trains these networks).
"""
import torch
import torch.nn as nn

from ..predictor import ops

DISC_LOGIT_INIT_SCALE = 1.0

_ACTIVATIONS = {"relu": nn.ReLU, "tanh": nn.Tanh, "sigmoid": nn.Sigmoid, "elu": nn.ELU, "selu": nn.SELU,
                "silu": nn.SiLU, "gelu": nn.GELU, "softplus": nn.Softplus, "None": nn.Identity, None: nn.Identity}


def _initializer(cfg):
    """network_builder.py:48-60 init factory (the subset the shipped configs name)."""
    name = cfg.get("name", "default")
    kw = {k: v for k, v in cfg.items() if k != "name"}
    table = {"default": lambda t: t, "const_initializer": lambda t: nn.init.constant_(t, **kw),
             "orthogonal_initializer": lambda t: nn.init.orthogonal_(t, **kw), "orthogonal": lambda t: nn.init.orthogonal_(t, **kw),
             "glorot_normal_initializer": lambda t: nn.init.xavier_normal_(t, **kw),
             "glorot_uniform_initializer": lambda t: nn.init.xavier_uniform_(t, **kw),
             "random_uniform_initializer": lambda t: nn.init.uniform_(t, **kw),
             "kaiming_normal": lambda t: nn.init.kaiming_normal_(t, **kw)}
    if name not in table:
        raise ValueError(f"initializer '{name}' is not supported")
    return table[name]


class FusedMLP(nn.Sequential):
    """nn.Sequential of [Linear, activation]* (network_builder.py:91-110) whose Linear+ReLU pairs run as one GEMM launch."""

    def forward(self, x):
        mods = list(self)
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, nn.Linear):
                relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
                x = ops.linear(x, m.weight, m.bias, relu=relu)
                i += 2 if relu else 1
            else:
                x = m(x)
                i += 1
        return x


def build_mlp(input_size, units, activation):
    layers, in_size = [], input_size
    for unit in units:
        layers.append(nn.Linear(in_size, unit))
        layers.append(_ACTIVATIONS[activation]())
        in_size = unit
    return FusedMLP(*layers)


class _Head(nn.Linear):
    def forward(self, x):
        return ops.linear(x, self.weight, self.bias, relu=False)


class AMPSeptBuilder:
    def __init__(self, **kwargs):
        self.params = None

    def load(self, params):
        self.params = params

    def build(self, name, **kwargs):
        return AMPSeptBuilder.Network(self.params, **kwargs)

    def __call__(self, name, **kwargs):
        return self.build(name, **kwargs)

    class Network(nn.Module):
        def __init__(self, params, **kwargs):
            super().__init__()
            self.self_obs_size = kwargs["self_obs_size"]
            self.task_obs_size = kwargs["task_obs_size"]
            self.task_obs_size_detail = kwargs["task_obs_size_detail"]
            if "people" in self.task_obs_size_detail:
                raise NotImplementedError("the 'people' point-net branch (amp_network_sept_builder.py:52-65) is not built")
            self.load(params)
            actions_num = kwargs["actions_num"]
            self.value_size = kwargs.get("value_size", 1)
            in_size = self.self_obs_size + self._task_units[-1]          # amp_network_sept_builder.py:35
            out_size = self.units[-1] if len(self.units) else in_size
            self.actor_cnn = nn.Sequential()
            self.critic_cnn = nn.Sequential()
            self.actor_mlp = build_mlp(in_size, self.units, self.activation)
            self.critic_mlp = build_mlp(in_size, self.units, self.activation) if self.separate else nn.Sequential()
            self.value = _Head(out_size, self.value_size)
            self.value_act = _ACTIVATIONS[self.value_activation]()
            self.mu = _Head(out_size, actions_num)
            self.mu_act = _ACTIVATIONS[self.space_config["mu_activation"]]()
            self.sigma_act = _ACTIVATIONS[self.space_config["sigma_activation"]]()
            if not self.space_config["fixed_sigma"]:
                raise NotImplementedError("state-dependent sigma is not built (the shipped config fixes it)")
            mlp_init = _initializer(self.initializer)
            for m in self.modules():                                      # network_builder.py:262-270
                if isinstance(m, nn.Linear):
                    mlp_init(m.weight)
                    if m.bias is not None:
                        nn.init.zeros_(m.bias)
            _initializer(self.space_config["mu_init"])(self.mu.weight)
            # amp_network_builder.py:20-25: with learn_sigma False sigma is a frozen parameter
            learn = self.space_config.get("learn_sigma", True)
            self.sigma = nn.Parameter(torch.zeros(actions_num, dtype=torch.float32), requires_grad=learn)
            with torch.no_grad():
                _initializer(self.space_config["sigma_init"])(self.sigma)
            self._build_disc(kwargs.get("amp_input_shape"))
            self.running_mean = kwargs["mean_std"].running_mean if kwargs.get("mean_std") is not None else None
            self.running_var = kwargs["mean_std"].running_var if kwargs.get("mean_std") is not None else None
            self._build_task_mlp()

        def load(self, params):
            self.separate = params["separate"]
            self.units = params["mlp"]["units"]
            self.activation = params["mlp"]["activation"]
            self.initializer = params["mlp"]["initializer"]
            if params["mlp"].get("d2rl", False):
                raise NotImplementedError("d2rl MLPs are not built")
            if "continuous" not in params["space"]:
                raise NotImplementedError("only the continuous action space is built")
            self.space_config = params["space"]["continuous"]
            self.value_activation = params.get("value_activation", "None")
            self._disc_units = params["disc"]["units"]
            self._disc_activation = params["disc"]["activation"]
            self._disc_initializer = params["disc"]["initializer"]
            self._task_units = params["task_mlp"]["units"]
            self._task_activation = params["task_mlp"]["activation"]
            self._task_initializer = params["task_mlp"]["initializer"]

        def is_separate_critic(self):
            return False

        def is_rnn(self):
            return False

        def get_default_rnn_state(self):
            return None

        # ------------------------------------------------------------------ forward pieces
        def forward(self, obs_dict):
            obs = obs_dict["obs"]
            states = obs_dict.get("rnn_states", None)
            return self.eval_actor(obs) + (self.eval_critic(obs), states)

        def eval_task(self, task_obs):
            return self._task_mlp(task_obs)

        def _split(self, obs):
            assert obs.shape[-1] == self.self_obs_size + self.task_obs_size
            return obs[:, :self.self_obs_size], obs[:, self.self_obs_size:self.self_obs_size + self.task_obs_size]

        def eval_critic(self, obs):
            self_obs, task_obs = self._split(obs)
            c_input = torch.cat([self_obs, self.eval_task(task_obs)], dim=-1)
            return self.value_act(self.value(self.critic_mlp(c_input)))

        def eval_actor(self, obs):
            self_obs, task_obs = self._split(obs)
            actor_input = torch.cat([self_obs, self.eval_task(task_obs)], dim=-1)
            a_out = self.actor_mlp(actor_input)
            mu = self.mu_act(self.mu(a_out))
            sigma = mu * 0.0 + self.sigma_act(self.sigma)
            return mu, sigma

        def eval_disc(self, amp_obs):
            return self._disc_logits(self._disc_mlp(amp_obs))

        def get_disc_logit_weights(self):
            return torch.flatten(self._disc_logits.weight)

        def get_disc_weights(self):
            weights = [torch.flatten(m.weight) for m in self._disc_mlp.modules() if isinstance(m, nn.Linear)]
            weights.append(torch.flatten(self._disc_logits.weight))
            return weights

        # ------------------------------------------------------------------ builders
        def _build_disc(self, input_shape):
            self._disc_mlp = build_mlp(input_shape[0], self._disc_units, self._disc_activation)
            self._disc_logits = _Head(self._disc_units[-1], 1)
            init = _initializer(self._disc_initializer)
            for m in self._disc_mlp.modules():
                if isinstance(m, nn.Linear):
                    init(m.weight)
                    nn.init.zeros_(m.bias)
            nn.init.uniform_(self._disc_logits.weight, -DISC_LOGIT_INIT_SCALE, DISC_LOGIT_INIT_SCALE)
            nn.init.zeros_(self._disc_logits.bias)

        def _build_task_mlp(self):
            detail = self.task_obs_size_detail
            assert "traj" in detail and "heightmap" in detail
            self._task_mlp = build_mlp(detail["traj"] + detail["heightmap"], self._task_units, self._task_activation)
            init = _initializer(self._task_initializer)
            for m in self._task_mlp.modules():
                if isinstance(m, nn.Linear):
                    init(m.weight)
                    nn.init.zeros_(m.bias)
