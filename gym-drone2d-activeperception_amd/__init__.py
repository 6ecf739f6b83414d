"""MI355X-native batched Drone2D active-perception step (hot path of smoggy-P/gym-Drone2D-ActivePerception).

Layout
  csrc/          HIP kernels + the C ABI of include/d2d.h (libd2d_hip.so)
  _abi.py        ctypes mirror of include/d2d.h
  _lib.py        loader of libd2d_hip.so (raises if the library is missing: there is no CPU fallback)
  params.py      the reference's Params / argparse surface (utils.py:65-171)
  host_init.py   world construction = the reference's __init__ (seeded, bit-identical)
  state.py       device-resident batch state
  vec_env.py     VecDrone2DEnv: B envs stepped in lock-step on one GPU
  env.py         Drone2DEnv2: single-env gym facade (gym-2d-perception-v2) over VecDrone2DEnv
  planners.py    --planner plugin surface (traj_planner.py): Primitive / NoMove = views of the device stages
  gaze.py        --gaze_method plugin surface (yaw_planner.py): Oxford = the device stage, NoControl, Rotating
  device_plugins.py  host tables + per-env state of the device planner / gaze stages (d2d_plan)
  sweeps.py      survivability sweeps (glob_survivability_calculator.py) on d2d_rollout
  runner.py      Experiment: one episode -> the reference's CSV row (experiment.py)
  dist.py        env sharding across GPUs, RCCL gather of episode statistics

The directory name is the one the build contract prescribes; because of the hyphens import it with
importlib.import_module('gym-drone2d-activeperception_amd') or through the alias module `drone2d_amd`.
"""
__version__ = '0.1.0'

from . import _abi            # noqa: F401
from .params import Params, with_defaults   # noqa: F401
