"""Batched ``ManEnv``: the reference's env API (reference environment/manenv.py:8-126,
environment/interface/environment.py:1-10) over the MI355X-native simulator.

Same constructor, methods, class attributes and defaults as the reference; the one
extension is ``n_envs`` (default 1).  With ``n_envs == 1`` every method returns what the
reference returns (``step() -> (ndarray(12,), bool)``, ``reset() -> float``); with
``n_envs > 1`` the same calls act on all envs in lockstep and return device tensors
(``[n,12]`` float64, ``[n]`` bool) / an ``ndarray[n]`` of stiffnesses.
"""
import numpy as np

from . import native
from .mjcf import load_model


class Env(object):
    """reference environment/interface/environment.py:1-10"""

    def __init__(self, sim_start, sim_step):
        self.sim_start = sim_start
        self.sim_step = sim_step

    def step(self, *args):
        raise NotImplementedError("Not implemented")

    def reset(self):
        raise NotImplementedError("Not implemented")


class SimulationError(Exception):
    """Counterpart of mujoco_py.builder.MujocoException (reference manenv.py:50)."""


class ManEnv(Env):
    # ids / names exactly as the reference (environment/manenv.py:12-18)
    joint_ids = list(range(11, 64))
    tendon_ids = list(range(1))
    finger_names = ['g12', 'g2']
    obj_name = 'OBJ'
    n_actuated = 2    # close_hand / loose_hand drive ctrl[0 .. n_actuated - 1] (`for i in range(2)`, reference manenv.py:93-101); 4 for the four-finger gripper

    def __init__(self, sim_start, sim_step, env_paths, is_vis=True, n_envs=1, device=0, contact_flag_mode="intent", check_scene=True,
                 tendon_damper="auto", joint_ids=None, tendon_ids=None, finger_names=None, n_actuated=None, max_cached_scenes=4):
        """``joint_ids`` / ``tendon_ids``: which model entries ``set_new_stiffness`` writes; default = the reference's class attributes
        (joints 11..63 and tendon 0: manenv.py:12-13), to be overridden for a scene with another layout (e.g. a smaller shell).
        ``tendon_damper``: how the damper of the composite's volume tendon is integrated (mjcf.load_model).
        "explicit" = MuJoCo's Euler step as restated; "implicit" = the rank-one implicit treatment; "auto" (default) = explicit,
        and a scene that fails the load-time check under it (the reference's soft ball / cylinder) is reloaded with "implicit",
        with a printed notice.  ``self.tendon_damper`` holds what the loaded scene runs with."""
        super().__init__(sim_start, sim_step)
        assert len(env_paths) > 0
        assert contact_flag_mode in ("intent", "reference")
        assert tendon_damper in ("auto", "explicit", "implicit")
        self._tendon_damper_arg = tendon_damper
        if joint_ids is not None:
            self.joint_ids = [int(j) for j in joint_ids]      # instance attributes shadow the class lists (which stay the reference's)
        if tendon_ids is not None:
            self.tendon_ids = [int(t) for t in tendon_ids]
        if finger_names is not None:      # e.g. ['g11', 'g12', 'g13', 'g2'] for the four-finger gripper (reference manenv.py:16)
            self.finger_names = list(finger_names)
        if n_actuated is not None:
            self.n_actuated = int(n_actuated)
        self.check_scene = check_scene
        self.max_cached_scenes = int(max_cached_scenes)   # scenes whose native batch stays allocated across load_env() calls (LRU; close() drops them)
        self.is_vis = is_vis  # no viewer exists; kept for signature parity (render() is a no-op)
        self.env_paths = env_paths
        self.n_envs = int(n_envs)
        self.device_index = device
        self.contact_flag_mode = contact_flag_mode
        self.rng = np.random  # the reference draws from the global NumPy RNG (manenv.py:104)
        self.n_resets = 0     # envs reset after a simulation warning (what `except MujocoException: self.reset()` did, manenv.py:50-51)
        self.n_capacity_resets = 0   # ... of which: envs that ran out of the KERNELS' contact capacity (SG_FLAG_CONTACTFULL: 64 per finger stream /
                                     # 128 per env) -- not a MuJoCo warning: the reference's nconmax is 500 (soft_grip_two_fingers.xml:8) and MuJoCo
                                     # would have carried on.  Counted apart so that a dataset job can refuse to paper over it (create_dataset)
        self._scenes = {}     # path -> the loaded scene (model, batch, buffers, the damper it runs with): load_env() of a scene seen before
                              # neither compiles, allocates nor dry-runs again (the reference's loop switches scene after EVERY episode)
        self._load(env_paths[0])
        self.is_closing = True

    # ---- model / batch management (reference manenv.py:27-41) ----
    _SCENE_STATE = ("model", "tendon_damper", "nmodel", "env", "_sens", "_flags", "_touch", "_ctrl", "_finger_bits", "_finger_bits_names", "_fingers_left")

    def _load(self, path, _damper=None):
        import torch
        if _damper is None and path in self._scenes:
            # A scene this instance has loaded before: its batch is kept alive, the load-time verdict (which damper it needs) with it.
            # What the reference's load_env leaves behind is a fresh MjSim -- the state after mj_resetData, the XML's own stiffness --
            # so: that state, no per-env stiffness, the reference-mode finger list refilled.
            self._scenes[path] = self._scenes.pop(path)      # most recently used last
            for k, v in self._scenes[path].items():
                setattr(self, k, v)
            self._ctrl[:] = 0
            self._sens.zero_()      # a fresh MjSim's sensordata and contact list are empty: get_sensor_sensordata() right after
            self._touch.zero_()     # load_env() must not show the previous episode's last sample
            self.stiffness = np.full(self.n_envs, np.nan)
            self._k_range = (300, 1400)
            self.env.set_stiffness(np.zeros(self.n_envs), [], [])
            self.env.reset(0, flags=self._flags)
            self._fingers_left.fill_((1 << len(self._finger_bits_names)) - 1)
            return
        want = _damper or self._tendon_damper_arg
        self.model = load_model(path, None if want == "auto" else want)
        self.tendon_damper = "implicit" if self.model.opt_implicit_tendon_damping else "explicit"
        if max(self.joint_ids) >= self.model.njnt or max(self.tendon_ids) >= self.model.ntendon:
            raise ValueError("scene %s has %d joints and %d tendons: the stiffness ids (joints %d..%d, tendons %s) do not fit it -- pass "
                             "joint_ids / tendon_ids" % (path, self.model.nv, self.model.ntendon, min(self.joint_ids), max(self.joint_ids),
                                                         self.tendon_ids))
        self.nmodel = native.NativeModel(self.model)
        if self.check_scene:
            # the load-time check runs BEFORE the batch is made, on a batch of ONE env: the dry run uses the model's own stiffness, so
            # every env of a batch would do exactly the same (r04 ran it on all n_envs and, for a scene that needs the implicit damper,
            # allocated two full batches: a third of the three-scene quick start's wall time, profiles/r05_dataset_end_to_end.txt)
            try:
                self._check_scene(path)
            except SimulationError as err:
                if want != "auto" or self.tendon_damper == "implicit":
                    raise
                print("NOTICE: %s\n        reloading it with tendon_damper=\"implicit\" (DESIGN.md D5)" % err)
                self._load(path, "implicit")
                return
        self.env = native.NativeBatch(self.nmodel, self.n_envs, self.device_index)
        dev = self.env.device
        self._sens = torch.zeros(self.n_envs, self.nmodel.nsensordata, dtype=torch.float64, device=dev)
        self._flags = torch.zeros(self.n_envs, dtype=torch.int32, device=dev)
        self._touch = torch.zeros(self.n_envs, dtype=torch.int32, device=dev)
        self._ctrl = np.zeros(self.nmodel.nu)
        self.stiffness = np.full(self.n_envs, np.nan)
        self._k_range = (300, 1400)  # range of the last set_new_stiffness draw: re-draws after a failure stay inside it
        # which finger boxes (bit 2*chain+box of `touch`) match each name in finger_names
        self._finger_bits = []
        bits = self._chain_geom_bits()
        for name in self.finger_names:
            v = sum(1 << b for b, gname in bits.items() if name in gname)
            self._finger_bits.append(v - (1 << 64) if v >= (1 << 63) else v)      # as a signed 64-bit pattern (torch has no uint64 ops)
        self._finger_bits_names = list(self.finger_names)
        # "reference" mode state: which names are still in the list the reference aliases and never refills (manenv.py:70,80), one
        # list per env, kept on the device as a bit mask (bit i = finger_names[i] not yet removed)
        self._fingers_left = torch.full((self.n_envs,), (1 << len(self._finger_bits_names)) - 1, dtype=torch.int32, device=dev)
        if self.check_scene:
            self.env.reset(0, flags=self._flags)   # the state after mj_resetData (a fresh MjSim's), as the check on the whole batch used to leave it
        self._scenes[path] = {k: getattr(self, k) for k in self._SCENE_STATE}
        while len(self._scenes) > max(1, self.max_cached_scenes):      # least recently used first; never the scene just loaded
            self._scenes.pop(next(iter(self._scenes)))

    def close(self):
        """drops every cached scene but the current one (a cached scene pins its native batch: state + work space, ~0.1 MB per env on
        the rows pipeline, ~0.5 MB per env on the tree pipeline -- hundreds of MB at 4096 envs)"""
        cur = [p for p, sc in self._scenes.items() if sc.get("env") is self.env]
        self._scenes = {p: self._scenes[p] for p in cur}

    def _check_scene(self, path, n_steps=40):
        """Fail loudly at load time for a scene that cannot produce data: the idle phase of an episode (reset + 40 env steps at
        ctrl = 0, the model's own stiffness) must run without a simulation warning.  The reference's soft ball / cylinder scenes
        start with the shell 0.14 / 0.30 deep inside the fingers (45 / 37 contacts at reset); with the volume tendon's damper
        integrated explicitly that start diverges within a few env steps (profiles/r02_ball_stability_probe.txt),
        and every reset starts there again -- the reference's `except MujocoException: self.reset()` (manenv.py:50-51) would
        loop forever."""
        import torch
        probe = native.NativeBatch(self.nmodel, 1, self.device_index)      # (one env: see _load)
        flags = torch.zeros(1, dtype=torch.int32, device=probe.device)
        bad = torch.zeros_like(flags)
        probe.reset(max(self.sim_start, 0), flags=flags)
        bad |= flags
        for _ in range(n_steps):
            probe.step(self.sim_step, flags=flags)
            bad |= flags
        if bool((bad != 0).any()):
            names = {1: "BADQPOS", 2: "BADQVEL", 4: "BADQACC", 8: "CONTACTFULL", 16: "CNSTRFULL", 32: "UNSUPPORTED_PAIR"}
            f = int(bad.max())
            raise SimulationError("scene %s does not survive its own idle phase with the %s tendon damper (flags %s within %d env steps "
                                  "at ctrl = 0); no dataset can be generated from it this way (check_scene=False loads it anyway)"
                                  % (path, self.tendon_damper, "|".join(v for k, v in names.items() if f & k), n_steps))

    def _chain_geom_bits(self):
        """bit index -> geom name for the moving finger boxes, in the kernels' order: (chain, body, geom) = geom id order (two boxes
        per finger in the two-finger class: bit 2 * chain + box)"""
        m = self.model
        moving = [g for g in range(m.ngeom) if m.body_weldid[m.geom_bodyid[g]] != 0 and m.geom_type[g] == 6]
        return {i: m.geom_names[g] or "" for i, g in enumerate(moving)}

    def _touch_bits(self):
        """per env the contact read-out as one integer: int32 word of sg_step for up to 32 finger boxes, else the words of
        sg_get_touch_words combined into an int64 (the four-finger gripper has 64 boxes)"""
        import torch
        if self.nmodel.nboxes <= 32:
            return self._touch
        w = self.env.touch_words(2).to(torch.int64) & 0xFFFFFFFF
        return w[:, 0] | (w[:, 1] << 32)

    def load_env(self, num):
        if num < len(self.env_paths):
            self._load(self.env_paths[num])
        else:
            print("Wrong number,")

    # ---- main methods ----
    def step(self, num_steps=-1, actions=None, min_dist=0.1):
        """reference manenv.py:44-53 (``actions`` / ``min_dist`` are unused there too)"""
        if num_steps < 1:
            num_steps = self.sim_step
        self.env.step(num_steps, sens=self._sens, flags=self._flags, touch=self._touch)
        bad = (self._flags != 0)
        if bool(bad.any()):  # mujoco_py raised -> the reference resets (re-drawing the stiffness) and carries on
            self.n_capacity_resets += int(((self._flags & native.SG_FLAG_CONTACTFULL) != 0).sum())
            self._reset_envs(bad)
        return self._result()

    def reset(self, range_min=300, range_max=1400):
        """reference manenv.py:55-63; the stiffness range is an extension (the reference always draws from U(300, 1400)):
        a rank of a sharded run passes its stiffness bin"""
        current_stiffness = self.set_new_stiffness(range_min, range_max)
        self.env.reset(max(self.sim_start, 0), sens=self._sens, flags=self._flags, touch=self._touch)
        self._ctrl[:] = 0  # mj_resetData clears ctrl
        bad = (self._flags != 0)
        if bool(bad.any()):
            self.n_capacity_resets += int(((self._flags & native.SG_FLAG_CONTACTFULL) != 0).sum())
            self._reset_envs(bad)
        return current_stiffness

    def _reset_envs(self, bad_mask, max_tries=5):
        """what `except MujocoException: self.reset()` does (reference manenv.py:50-51), for the flagged envs only: a new
        stiffness from the range of the last draw, mj_resetData (ctrl of those envs stays 0 until the next close / toggle, as in
        the reference), forward, sim_start steps.  Raises only when envs are still flagged after max_tries resets."""
        import torch
        lo, hi = self._k_range
        for _ in range(max_tries):
            idx = torch.nonzero(bad_mask).flatten().cpu().numpy()
            if idx.size == 0:
                return
            self.n_resets += int(idx.size)
            self.stiffness[idx] = self.rng.uniform(lo, hi, size=idx.size) if idx.size > 1 else self.rng.uniform(lo, hi)
            self.env.set_stiffness(self.stiffness, self.joint_ids, self.tendon_ids)
            mask = bad_mask.to(torch.uint8).contiguous()
            flags = torch.zeros_like(self._flags)
            self.env.reset(max(self.sim_start, 0), sens=self._sens, flags=flags, touch=self._touch, mask=mask)
            bad_mask = bad_mask & (flags != 0)
        idx = torch.nonzero(bad_mask).flatten().cpu().numpy()
        if idx.size:
            raise SimulationError("envs keep failing after %d resets: %s" % (max_tries, idx))

    def _contact_flags(self):
        import torch
        touch = self._touch_bits()
        if self.contact_flag_mode == "intent":
            ok = torch.ones(self.n_envs, dtype=torch.bool, device=touch.device)
            for bits in self._finger_bits:
                ok &= (touch & bits) != 0
            return ok
        # "reference": reproduce the aliased, never-refilled list of manenv.py:70-83 per env -- on the device, no host round trip:
        # a name leaves an env's list the first time one of its boxes touches an object geom; once the list is empty the flag is up
        # whenever there is any contact at all
        ncon = self.env.solver_stats()["ncon"]
        for i, bits in enumerate(self._finger_bits):
            self._fingers_left &= ~(((touch & bits) != 0).to(torch.int32) << i)   # (touch: int32, or int64 beyond 32 boxes)
        return (self._fingers_left == 0) & (ncon > 0)

    def _result(self):
        flag = self._contact_flags()
        if self.n_envs == 1:
            return self._sens[0].cpu().numpy().copy(), bool(flag[0])
        return self._sens.clone(), flag

    def get_sensor_sensordata(self):
        """reference manenv.py:65-85"""
        return self._result()

    def toggle_grip(self):
        if self.is_closing:
            self.loose_hand()
        else:
            self.close_hand()

    def close_hand(self):
        self._ctrl[:self.n_actuated] = -0.2
        self.env.set_ctrl_broadcast(self._ctrl)
        self.is_closing = True

    def loose_hand(self):
        self._ctrl[:self.n_actuated] = 0.2
        self.env.set_ctrl_broadcast(self._ctrl)
        self.is_closing = False

    def set_new_stiffness(self, range_min=300, range_max=1400):
        """reference manenv.py:103-109: one draw per env from the global NumPy RNG, written to
        jnt_stiffness[joint_ids] and tendon_stiffness[tendon_ids]"""
        self._k_range = (range_min, range_max)
        if self.n_envs == 1:
            new_value = self.rng.uniform(range_min, range_max)
            self.stiffness[0] = new_value
        else:
            new_value = self.rng.uniform(range_min, range_max, size=self.n_envs)
            self.stiffness[:] = new_value
        self.env.set_stiffness(self.stiffness, self.joint_ids, self.tendon_ids)
        return new_value

    def set_stiffness_values(self, values):
        """batched extension: explicit per-env stiffness (e.g. a stiffness-bin sweep)"""
        self.stiffness[:] = np.asarray(values, dtype=np.float64).reshape(self.n_envs)
        self.env.set_stiffness(self.stiffness, self.joint_ids, self.tendon_ids)

    def get_env(self):
        return self.env

    def render(self):
        pass  # viewer is out of scope

    # ---- fused episode: the create_dataset.py schedule without a host round trip per step ----
    def rollout(self, schedule, out=None, reset=True):
        """Runs ``len(schedule)`` env steps; ``schedule[t]`` is the broadcast ctrl (or None = unchanged) applied
        before step t.  Writes sensordata into ``out[n, T, 12]`` (allocated when None) and returns
        ``(out, flags_or[n])``.  Envs that fail are NOT reset here (their flags are returned)."""
        import torch
        T = len(schedule)
        dev = self.env.device
        nsd = self.nmodel.nsensordata
        if out is None:
            out = torch.empty(self.n_envs, T, nsd, dtype=torch.float64, device=dev)
        assert out.shape == (self.n_envs, T, nsd) and out.is_contiguous()
        flags_or = torch.zeros(self.n_envs, dtype=torch.int32, device=dev)
        if reset:
            self.env.reset(max(self.sim_start, 0), sens=self._sens, flags=self._flags, touch=self._touch)
            self._ctrl[:] = 0
            flags_or |= self._flags
        for t in range(T):
            if schedule[t] is not None:
                self._ctrl[:] = schedule[t]
                self.env.set_ctrl_broadcast(self._ctrl)
            self.env.step(self.sim_step, sens=out[:, t], sens_stride=T * nsd, flags=self._flags, touch=self._touch)
            flags_or |= self._flags
        return out, flags_or

    # specs
    @staticmethod
    def get_std_spec(args):
        spec = {
            "sim_start": args.sim_start,
            "sim_step": args.sim_step,
            "env_paths": args.mujoco_model_paths,
            "is_vis": args.vis,
        }
        for extra in ("n_envs", "device", "contact_flag_mode", "check_scene", "tendon_damper", "joint_ids", "tendon_ids", "finger_names", "n_actuated"):
            if hasattr(args, extra):
                spec[extra] = getattr(args, extra)
        return spec
