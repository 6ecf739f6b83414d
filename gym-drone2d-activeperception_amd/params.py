"""Config surface of the reference, kept verbatim: utils.py:65-171 (`Params`, `Params.from_parser`).

Same attribute names, same defaults, same quirks (`--debug` is store_false so render defaults to True;
`init_pos` is stored as `init_position`).  The env also accepts any attribute-style object (EasyDict in
script/train.py:16) and fills attributes that object lacks from these defaults (`with_defaults`)."""
import argparse
import copy

_DEFAULTS = dict(
    env='gym-2d-perception-v2', debug=True, record_img=False, trained_policy=False,
    policy_dir='./trained_policy/lookahead.zip', dt=0.1, map_scale=10, map_size=[500, 500], agent_radius=10,
    drone_max_acceleration=40, drone_radius=10, drone_max_yaw_speed=80, drone_view_depth=80,
    drone_view_range=90, img_dir='./', max_flight_time=80, gaze_method='LookAhead', planner='Primitive',
    var_cam=0, drone_max_speed=40, motion_profile='CVM', pillar_number=0, agent_number=10,
    agent_max_speed=40, map_id=0, init_pos=[50, 50], target_list=[[50, 460]],
    static_map='maps/empty_map.npy')


class Params:
    def __init__(self, **kw):
        unknown = set(kw) - set(_DEFAULTS)
        if unknown:
            raise TypeError(f'Params got unexpected arguments {sorted(unknown)}')
        v = dict(_DEFAULTS)
        v.update(kw)
        debug = v.pop('debug')
        self.render = bool(debug)       # utils.py:75-80
        self.record = not debug
        self.init_position = v.pop('init_pos')   # utils.py:104
        for k, val in v.items():
            setattr(self, k, copy.copy(val) if isinstance(val, list) else val)

    @classmethod
    def from_parser(cls, argv=None):
        ap = argparse.ArgumentParser(description='Initialize Params class with command-line arguments')
        ap.add_argument('--env', default=_DEFAULTS['env'])
        ap.add_argument('--debug', action='store_false')
        ap.add_argument('--record_img', action='store_true')
        ap.add_argument('--trained_policy', action='store_true')
        ap.add_argument('--policy_dir', default=_DEFAULTS['policy_dir'])
        ap.add_argument('--dt', type=float, default=0.1)
        ap.add_argument('--map_size', nargs=2, type=int, default=[500, 500])
        ap.add_argument('--init_pos', nargs=2, type=int, default=[50, 50])
        ap.add_argument('--target_list', nargs='+', type=int, default=[[50, 460]])
        ap.add_argument('--img_dir', default='./')
        ap.add_argument('--gaze_method', default='LookAhead')
        ap.add_argument('--planner', default='Primitive')
        ap.add_argument('--motion_profile', default='CVM')
        ap.add_argument('--static_map', default='maps/empty_map.npy')
        for name in ('map_scale', 'agent_radius', 'drone_max_acceleration', 'drone_radius',
                     'drone_max_yaw_speed', 'drone_view_depth', 'drone_view_range', 'max_flight_time',
                     'var_cam', 'drone_max_speed', 'pillar_number', 'agent_number', 'agent_max_speed',
                     'map_id'):
            ap.add_argument('--' + name, type=int, default=_DEFAULTS[name])
        a = ap.parse_args(argv)
        return cls(**vars(a))


class ParamsView:
    """Plain attribute bag returned by with_defaults()."""


def with_defaults(params):
    """Attribute-style view of `params` that falls back to the reference defaults for missing fields
    (script/train.py builds an EasyDict without planner / map_id / max_flight_time ...)."""
    v = ParamsView()
    d = Params()
    for k, val in vars(d).items():
        setattr(v, k, val)
    src = params if isinstance(params, dict) else {k: getattr(params, k) for k in dir(params)
                                                   if not k.startswith('_') and not callable(getattr(params, k))}
    for k, val in src.items():
        setattr(v, k, val)
    if 'init_pos' in src and 'init_position' not in src:
        v.init_position = src['init_pos']
    return v
