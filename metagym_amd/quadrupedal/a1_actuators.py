"""`A1Actuators` — N robots' worth of what `a1.A1` / `minitaur.Minitaur` / `LaikagoMotorModel` do around the physics
step (metagym/quadrupedal/robots/a1.py, minitaur.py, laikago_motor.py), with the reference's method names.

    act = A1Actuators(num_envs=4096, device="cuda:0")               # POSITION mode, kp/kd of a1.py:63-68
    act.Reset()                                                     # Minitaur.Reset: history cleared
    act.ReceiveObservation(q, qd, base_quat, rpy_rate)              # first observation (Reset -> _SettleDownForReset,
                                                                    # minitaur.py:460-462). A bare `a1.A1(...)` observes a second
                                                                    # time in __init__ (:226): call it twice to mirror that
    torques = act.Step(action, physics)                             # 13 x (ApplyAction -> physics -> ReceiveObservation)

`physics(torques) -> (q, qd, base_quat, rpy_rate)` is the caller's simulator: the A1 body is not part of this package
(a1.urdf and PyBullet are absent from the reference tree). All tensors are float64 `[num_envs, k]` on the device; the
kernels behind the C ABI (metagym_amd/csrc/a1.hip) keep them as `[k][num_envs]`. No CPU fallback."""
import ctypes as C
import enum

import numpy as np
import torch

from .. import _lib

NUM_MOTORS = _lib.A1_NUM_MOTORS
MOTOR_NAMES = ["%s_%s_joint" % (leg, part) for leg in ("FR", "FL", "RR", "RL") for part in ("hip", "upper", "lower")]   # a1.py:27-40
INIT_MOTOR_ANGLES = np.array([0, 0.9, -1.8] * 4, dtype=np.float64)          # a1.py:71
# a1.py:63-68: abduction / hip / knee gains
DEFAULT_KP = [80.0, 80.0, 80.0] * 4
DEFAULT_KD = [1.0, 2.0, 2.0] * 4


class MotorControlMode(enum.Enum):
    """robots/robot_config.py:13-27 (PWM is Minitaur-only and rejected like laikago_motor.py:119-121)."""
    POSITION = 1
    TORQUE = 2
    HYBRID = 3


class SoA(object):
    """Explicit layout marker: `SoA(t)` declares that the float64 device tensor `t` is already `[k][num_envs]` (component-major,
    the kernels' layout) — what a batched simulator naturally holds. Anything not wrapped is `[num_envs, k]`."""
    __slots__ = ("t",)

    def __init__(self, t):
        self.t = t.t if isinstance(t, SoA) else t


class A1Actuators(object):
    def __init__(self, num_envs, device="cuda:0", time_step=0.002, action_repeat=13, control_latency=0.002,
                 pd_latency=0.0, motor_control_mode=MotorControlMode.POSITION, motor_kp=DEFAULT_KP, motor_kd=DEFAULT_KD,
                 motor_torque_limits=33.5, enable_action_interpolation=False, enable_clip_motor_commands=False,
                 enable_action_filter=False, history_len=100):
        self.device = _lib.canonical_device(device)
        if self.device.type != "cuda":
            raise _lib.MetaGymHipError("metagym_amd has no CPU path: device must be a ROCm GPU, got %r" % (device,))
        self._lib = _lib.load()
        self.num_envs = N = int(num_envs)
        self.num_motors = NUM_MOTORS
        self.time_step = float(time_step)
        self._action_repeat = int(action_repeat)
        self._enable_action_interpolation = bool(enable_action_interpolation)
        self._action_filter = None
        if enable_action_filter:                            # Minitaur._BuildActionFilter minitaur.py:1438-1443
            from .a1_wrappers import ActionFilter
            self._action_filter = ActionFilter.butter(num_envs, 1 / (float(time_step) * int(action_repeat)), device)
        self._last_action = None
        self._step_counter = 0
        f64 = dict(dtype=torch.float64, device=self.device)
        self._history = torch.zeros(int(history_len), _lib.A1_OBS_DIM, N, **f64)
        self._count = torch.zeros(N, dtype=torch.int32, device=self.device)
        self._head = torch.zeros(N, dtype=torch.int32, device=self.device)
        self._observed_torque = torch.zeros(NUM_MOTORS, N, **f64)
        self._control_obs = torch.zeros(_lib.A1_OBS_DIM, N, **f64)
        self._torque = torch.zeros(NUM_MOTORS, N, **f64)
        self._keep = {}
        c = self._cfg = _lib.A1ActuatorConfig()
        c.time_step, c.action_repeat, c.history_len = self.time_step, self._action_repeat, int(history_len)
        c.clip_commands, c.max_angle_change = int(bool(enable_clip_motor_commands)), 0.2
        self._set_mode(motor_control_mode)
        self.SetControlLatency(control_latency)
        self.SetPDLatency(pd_latency)
        self.SetMotorGains(motor_kp, motor_kd)
        self.SetMotorStrengthRatios(np.full(NUM_MOTORS, 1.0))
        if motor_torque_limits is None:
            c.has_torque_limit = 0
        else:
            c.has_torque_limit = 1
            c.torque_limit[:] = list(np.broadcast_to(np.asarray(motor_torque_limits, dtype=np.float64), (NUM_MOTORS,)))
        s = self._st = _lib.A1ActuatorState()
        s.history, s.count, s.head = self._history.data_ptr(), self._count.data_ptr(), self._head.data_ptr()
        s.observed_torque, s.control_obs = self._observed_torque.data_ptr(), self._control_obs.data_ptr()

    # ---- configuration (minitaur.py:1246-1328) ------------------------------------------------------
    def _set_mode(self, mode):
        if not isinstance(mode, MotorControlMode):
            mode = MotorControlMode(getattr(mode, "value", mode))
        self._motor_control_mode = mode
        self._cfg.mode = mode.value

    def _per_env(self, key, value, shape):
        """scalar / per-motor values go into the config struct, [N]-shaped tensors stay on the device."""
        if torch.is_tensor(value) and value.dim() >= 1 and value.shape[0] == self.num_envs and value.dim() == len(shape) + 1:
            t = value.to(self.device, torch.float64)
            t = t.t().contiguous() if t.dim() == 2 else t.contiguous()
            self._keep[key] = t
            return t
        self._keep.pop(key, None)
        return None

    def SetControlLatency(self, latency):
        t = self._per_env("control_latency", latency, ())
        self._cfg.control_latency_env = t.data_ptr() if t is not None else None
        if t is None:
            self._cfg.control_latency = float(latency)

    def SetPDLatency(self, latency):
        t = self._per_env("pd_latency", latency, ())
        self._cfg.pd_latency_env = t.data_ptr() if t is not None else None
        if t is None:
            self._cfg.pd_latency = float(latency)

    def SetMotorGains(self, kp, kd):
        """kp, kd: scalar, 12 values, or `[num_envs, 12]` tensors (locomotion_gym_env.py:388-392 draws them per robot)."""
        for name, v in (("kp", kp), ("kd", kd)):
            t = self._per_env(name, v, (NUM_MOTORS,))
            setattr(self._cfg, name + "_env", t.data_ptr() if t is not None else None)
            if t is None:
                getattr(self._cfg, name)[:] = list(np.broadcast_to(np.asarray(v, dtype=np.float64), (NUM_MOTORS,)))

    def SetMotorStrengthRatios(self, ratios):
        self._cfg.strength[:] = list(np.broadcast_to(np.asarray(ratios, dtype=np.float64), (NUM_MOTORS,)))

    def SetMotorStrengthRatio(self, ratio):
        self.SetMotorStrengthRatios(np.full(NUM_MOTORS, ratio))

    # ---- the sub-step (minitaur.py:232-255) ------------------------------------------------------------
    def _soa(self, x, k):
        """`[num_envs, k]` -> the kernels' `[k][num_envs]` layout (one transpose-copy). The layout is never guessed from the
        shape (a `[k, num_envs]` tensor is indistinguishable from `[num_envs, k]` when num_envs == k): a simulator that already
        holds its state as `[k][num_envs]` says so by wrapping the tensor in `SoA(...)`, which is then taken without a copy."""
        if isinstance(x, SoA):
            t = x.t
            if not (t.dtype == torch.float64 and t.device == self.device and tuple(t.shape) == (k, self.num_envs) and t.is_contiguous()):
                raise ValueError("SoA tensor must be contiguous float64 [%d, num_envs=%d] on %s, got %s %s on %s"
                                 % (k, self.num_envs, self.device, t.dtype, tuple(t.shape), t.device))
            return t
        x = torch.as_tensor(x, dtype=torch.float64, device=self.device)
        if tuple(x.shape) != (self.num_envs, k):
            raise ValueError("expected [num_envs=%d, %d], got %s (wrap a [k, num_envs] tensor in metagym_amd.quadrupedal.SoA)"
                             % (self.num_envs, k, tuple(x.shape)))
        return x.t().contiguous()

    def Reset(self, mask=None):
        """Minitaur.Reset (minitaur.py:434-441): history cleared, counters zeroed; the next ReceiveObservation starts it."""
        if mask is None:
            self._count.zero_()
            self._observed_torque.zero_()
        else:
            m = torch.as_tensor(mask, device=self.device).bool()
            self._count.masked_fill_(m, 0)                                 # (in place, one launch each, nothing syncs with the host)
            self._observed_torque.masked_fill_(m, 0.0)
        if mask is None:                # the host-side scalars describe the batch as a whole: a partial reset leaves them
            self._step_counter = 0
            self._last_action = None
        if self._action_filter is not None:                 # _ResetActionFilter minitaur.py:1445-1446
            self._action_filter.reset(mask)

    def ProcessAction(self, action, substep_count):
        """minitaur.py:1419-1436 — returns (command, last_command or None, lerp) for the kernel to combine."""
        if self._enable_action_interpolation and self._last_action is not None:
            return action, self._last_action, float(substep_count + 1) / self._action_repeat
        return action, None, 0.0

    def ApplyAction(self, motor_commands, motor_control_mode=None):
        """-> torques `[num_envs, 12]` (what _SetMotorTorqueByIds hands to the physics)."""
        if motor_control_mode is not None:
            self._set_mode(motor_control_mode)
        k = 5 * NUM_MOTORS if self._motor_control_mode is MotorControlMode.HYBRID else NUM_MOTORS
        return self._apply(self._soa(motor_commands, k), None, 0.0)

    def _apply(self, cmd, last, lerp):
        with torch.cuda.device(self.device):
            rc = self._lib.mg_a1_apply_action(C.byref(self._cfg), self.num_envs, C.byref(self._st), _lib.ptr(cmd),
                                              _lib.ptr(last), float(lerp), _lib.ptr(self._torque),
                                              _lib.current_stream(self.device))
        _lib.check(rc, "mg_a1_apply_action")
        return self._torque.t()

    def _receive_and_apply(self, motor_angles, motor_velocities, base_orientation, base_rpy_rate, cmd, last, lerp):
        q, qd = self._soa(motor_angles, NUM_MOTORS), self._soa(motor_velocities, NUM_MOTORS)
        quat, rate = self._soa(base_orientation, 4), self._soa(base_rpy_rate, 3)
        with torch.cuda.device(self.device):
            rc = self._lib.mg_a1_receive_and_apply(C.byref(self._cfg), self.num_envs, C.byref(self._st), _lib.ptr(q), _lib.ptr(qd),
                                                   _lib.ptr(quat), _lib.ptr(rate), _lib.ptr(cmd), _lib.ptr(last), float(lerp),
                                                   _lib.ptr(self._torque), _lib.current_stream(self.device))
        _lib.check(rc, "mg_a1_receive_and_apply")
        return self._torque.t()

    def ReceiveObservation(self, motor_angles, motor_velocities, base_orientation, base_rpy_rate, clear_mask=None, only_mask=None):
        """The four arguments are what the reference reads from Bullet at this point (getJointStates, base orientation
        relative to the initial one, angular velocity in the body frame; minitaur.py:1190-1200, :840-872). `only_mask`: the
        first observation after Reset(mask) of PART of the batch — every other robot is left exactly as it is."""
        q, qd = self._soa(motor_angles, NUM_MOTORS), self._soa(motor_velocities, NUM_MOTORS)
        quat, rate = self._soa(base_orientation, 4), self._soa(base_rpy_rate, 3)
        cm = None if clear_mask is None else torch.as_tensor(clear_mask, device=self.device).to(torch.uint8).contiguous()
        if only_mask is not None:       # kernel mask values: 0 push, 1 clear + push, 2 untouched
            if only_mask.dtype == torch.uint8:      # the caller's ready-made code array (1 = this robot, 2 = untouched)
                cm = only_mask
            else:
                om = torch.as_tensor(only_mask, device=self.device).bool()
                cm = torch.where(om, torch.ones_like(om, dtype=torch.uint8), torch.full_like(om, 2, dtype=torch.uint8)).contiguous()
        with torch.cuda.device(self.device):
            rc = self._lib.mg_a1_receive_observation(C.byref(self._cfg), self.num_envs, C.byref(self._st), _lib.ptr(q),
                                                     _lib.ptr(qd), _lib.ptr(quat), _lib.ptr(rate), _lib.ptr(cm),
                                                     _lib.current_stream(self.device))
        _lib.check(rc, "mg_a1_receive_observation")

    def Step(self, action, physics, control_mode=None, filter_init_mask=None):
        """Minitaur.Step: `action_repeat` x (ProcessAction, ApplyAction, physics(torques), ReceiveObservation).
        Returns the applied torques `[action_repeat, num_envs, 12]`."""
        if control_mode is not None:
            self._set_mode(control_mode)
        k = 5 * NUM_MOTORS if self._motor_control_mode is MotorControlMode.HYBRID else NUM_MOTORS
        if self._action_filter is not None:                 # _FilterAction minitaur.py:1448-1457
            if self._step_counter == 0:
                self._action_filter.init_history(self.GetMotorAngles())
            elif filter_init_mask is not None:              # first step of the robots that were just reset
                self._action_filter.init_history(self.GetMotorAngles(), filter_init_mask)
            action = self._action_filter.filter(action)
        act = self._soa(action, k)
        torques = []
        cmd, last, lerp = self.ProcessAction(act, 0)
        t = self._apply(cmd, last, lerp)
        for i in range(self._action_repeat):
            torques.append(t.clone())
            q, qd, quat, rate = physics(t)
            if i + 1 < self._action_repeat:      # ReceiveObservation of sub-step i and ApplyAction of i + 1 in one launch
                cmd, last, lerp = self.ProcessAction(act, i + 1)
                t = self._receive_and_apply(q, qd, quat, rate, cmd, last, lerp)
            else:
                self.ReceiveObservation(q, qd, quat, rate)
            self._step_counter += 1
        self._last_action = act
        return torch.stack(torques)

    def StepFused(self, action, fused_physics, filter_init_mask=None):
        """Minitaur.Step when the physics can run the whole action-repeat loop in one launch with the PD motor model evaluated
        inside it (`fused_physics(command_soa, actuators) -> log`, float64 `[action_repeat, 43, num_envs]`: one true
        observation per sub-step, torques included — e.g. WalkerBatchEnv.step_actuated). All three motor modes (POSITION with
        shared or per-robot gains, HYBRID, TORQUE), pd latency 0, no command clip, no interpolation (the reference's A1
        configuration, env_builder.py:44-52); any control latency. The log is pushed on the history in one launch
        (mg_a1_receive_log). Returns the applied torques `[action_repeat, num_envs, 12]`."""
        c = self._cfg
        if not self.can_fuse():
            raise _lib.MetaGymHipError("StepFused needs pd latency 0 and no command clip / interpolation (the motor model is evaluated "
                                       "on the CURRENT joint state inside the physics launch); use Step")
        if self._action_filter is not None:
            if self._step_counter == 0:
                self._action_filter.init_history(self.GetMotorAngles())
            elif filter_init_mask is not None:
                self._action_filter.init_history(self.GetMotorAngles(), filter_init_mask)
            action = self._action_filter.filter(action)
        act = self._soa(action, 5 * NUM_MOTORS if self._motor_control_mode is MotorControlMode.HYBRID else NUM_MOTORS)
        log = fused_physics(act, self)
        assert log.shape == (self._action_repeat, _lib.A1_OBS_DIM, self.num_envs) and log.is_contiguous()
        with torch.cuda.device(self.device):
            rc = self._lib.mg_a1_receive_log(C.byref(c), self.num_envs, C.byref(self._st), _lib.ptr(log), self._action_repeat,
                                             _lib.current_stream(self.device))
        _lib.check(rc, "mg_a1_receive_log")
        self._step_counter += self._action_repeat
        self._last_action = act
        return log[:, 2 * NUM_MOTORS:3 * NUM_MOTORS, :].permute(0, 2, 1)

    def can_fuse(self):
        """StepFused's preconditions: the motor model runs on the CURRENT joint state inside the physics launch."""
        c = self._cfg
        return not (c.pd_latency != 0.0 or c.pd_latency_env or c.clip_commands or self._enable_action_interpolation)

    def fused_spec(self):
        """What a physics needs to evaluate this motor model itself (WalkerBatchEnv.step_actuated's keyword arguments)."""
        c = self._cfg
        kp, kd, strength, limit = self.motor_model_parameters()
        return dict(kp=kp, kd=kd, strength=strength, limit=limit,
                    mode={MotorControlMode.POSITION: "position", MotorControlMode.HYBRID: "hybrid", MotorControlMode.TORQUE: "torque"}[self._motor_control_mode],
                    kp_env=self._keep.get("kp") if c.kp_env else None, kd_env=self._keep.get("kd") if c.kd_env else None)

    def motor_model_parameters(self):
        """(kp, kd, strength, torque limit) per motor, for a physics that evaluates the PD model itself."""
        c = self._cfg
        lim = list(c.torque_limit) if c.has_torque_limit else [1e30] * NUM_MOTORS
        return list(c.kp), list(c.kd), list(c.strength), lim

    def GetTimeSinceReset(self):
        return self._step_counter * self.time_step

    # ---- sensor getters (minitaur.py:755-885; the noise standard deviations are zero, :48) -----------------------
    def _sensors(self, **want):
        outs = {}
        f64 = dict(dtype=torch.float64, device=self.device)
        shapes = dict(angles=(NUM_MOTORS, self.num_envs), vels=(NUM_MOTORS, self.num_envs),
                      torques=(NUM_MOTORS, self.num_envs), rate=(3, self.num_envs), energy=(self.num_envs,))
        for k in shapes:
            outs[k] = torch.empty(*shapes[k], **f64) if want.get(k) else None
        with torch.cuda.device(self.device):
            rc = self._lib.mg_a1_sensors(C.byref(self._cfg), self.num_envs, C.byref(self._st), _lib.ptr(outs["angles"]),
                                         _lib.ptr(outs["vels"]), _lib.ptr(outs["torques"]), _lib.ptr(outs["rate"]),
                                         _lib.ptr(outs["energy"]), _lib.current_stream(self.device))
        _lib.check(rc, "mg_a1_sensors")
        return outs

    def GetMotorAngles(self):
        return self._sensors(angles=True)["angles"].t()

    def GetMotorVelocities(self):
        return self._sensors(vels=True)["vels"].t()

    def GetMotorTorques(self):
        return self._sensors(torques=True)["torques"].t()

    def GetBaseRollPitchYawRate(self):
        return self._sensors(rate=True)["rate"].t()

    def GetEnergyConsumptionPerControlStep(self):
        return self._sensors(energy=True)["energy"]

    def GetControlObservation(self):
        """`_control_observation` (minitaur.py:1234-1237): `[num_envs, 43]`."""
        return self._control_obs.t()

    def GetTrueMotorTorques(self):
        return self._observed_torque.t()

    def state_dict(self):
        return dict(history=self._history.clone(), count=self._count.clone(), head=self._head.clone(),
                    observed_torque=self._observed_torque.clone(), control_obs=self._control_obs.clone(),
                    step_counter=self._step_counter,
                    last_action=None if self._last_action is None else self._last_action.clone())

    def load_state_dict(self, sd):
        for k, t in (("history", self._history), ("count", self._count), ("head", self._head),
                     ("observed_torque", self._observed_torque), ("control_obs", self._control_obs)):
            assert sd[k].shape == t.shape, "state_dict[%s] has shape %s, expected %s" % (k, tuple(sd[k].shape), tuple(t.shape))
            t.copy_(sd[k])
        self._step_counter = int(sd["step_counter"])
        self._last_action = None if sd["last_action"] is None else sd["last_action"].to(self.device).clone()
