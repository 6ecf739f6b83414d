"""--planner plugin surface (reference: traj_planner.py).

Interface kept from the reference so its callers and any third-party planner drop in:
    planner = planner_list[params.planner](drone, params)
    planner.set_target(xy) ; planner.plan(drone, dt) -> bool ; planner.replan_check(drone) -> (bool, swep_map)
    planner.trajectory (.positions / .velocities / .accelerations, len(), pop(), clear()) ; planner.target

`NoMove` also exists on the device (D2D_PLANNER_NOMOVE).  `Primitive` here is a host restatement of the
reference's motion-primitive A* (traj_planner.py:78-233): it runs between the two halves of the device
step (VecDrone2DEnv.perceive / act) and sees the drone's explored map and tracker estimates through the
same attribute names the reference planner reads (drone.map.get_grid, drone.trackers[k].active /
.estimate_pos / .radius).  MPC needs the proprietary FORCESPRO solver and Jerk_Primitive is not part of
this hot path; asking for them raises.
"""
import numpy as np
from numpy.linalg import norm


class Trajectory2D:
    """utils.py:280-298."""

    def __init__(self):
        self.positions, self.velocities, self.accelerations = [], [], []

    def pop(self):
        self.positions.pop(0)
        self.velocities.pop(0)
        self.accelerations.pop(0)

    def clear(self):
        self.positions, self.velocities, self.accelerations = [], [], []

    def __len__(self):
        return len(self.positions)


class Planner:
    """traj_planner.py:18-66."""

    def __init__(self, drone, params):
        self.trajectory = Trajectory2D()
        self.params = params
        self.target = np.array([drone.x, drone.y, 0, 0])

    def set_target(self, target):
        self.target = np.zeros(4)
        self.target[:2] = target

    def is_free(self, position, t, occupancy_map, trackers):
        """Five-probe static test at drone_radius + 10, then every active tracker's constant-velocity
        prediction at time t (traj_planner.py:28-59)."""
        if np.isnan(position).any():
            return False
        d = self.params.drone_radius + 10
        x, y = position[0], position[1]
        for qx, qy in ((x - d, y), (x, y), (x + d, y), (x, y - d), (x, y + d)):
            if occupancy_map.get_grid(qx, qy) == 1:
                return False
        for tr in trackers:
            if tr.active:
                if norm(position - tr.estimate_pos(t)) <= self.params.drone_radius + tr.radius + 5 + self.params.var_cam:
                    return False
        return True

    def plan(self, drone, update_t):
        raise NotImplementedError('No planner implemented!')

    def replan_check(self, drone):
        raise NotImplementedError('No replan checker implemented!')


class NoMove(Planner):
    """traj_planner.py:68-76."""

    def plan(self, drone, dt):
        self.target = np.array([-1, -1, 0, 0])
        return True

    def replan_check(self, drone):
        return False, drone.map.grid_map


class _Node:
    __slots__ = ('position', 'velocity', 'cost', 'total_cost', 'index', 'parent_index', 'coeff', 'itr')

    def __init__(self, pos, vel, cost, target, parent_index, coeff, itr):
        self.position, self.velocity, self.cost = pos, vel, cost
        self.parent_index, self.coeff, self.itr = parent_index, coeff, itr
        self.total_cost = cost + 0.5 * norm(pos - target) + 0.1 * norm(vel)       # traj_planner.py:88
        self.index = (round(pos[0]) // 10, round(pos[1]) // 10, round(vel[0]), round(vel[1]))   # :93


class Primitive(Planner):
    """Motion-primitive A* (traj_planner.py:78-233): 8 x 8 constant accelerations held for 2 s, at most
    99 expansions, 8 collision samples per primitive, trajectory re-sampled every dt."""

    def __init__(self, drone, params):
        super().__init__(drone, params)
        a, v = params.drone_max_acceleration, params.drone_max_speed
        self.u_space = np.arange(-a, a, 0.4 * v - 5) if v <= 40 else np.arange(-a, a, 4)      # :98-101
        self.dt = 2
        self.sample_num = v * self.dt // params.map_scale                                    # :104
        self.target = np.array([drone.x, drone.y, 0, 0])
        self.search_threshold = 10
        self.phi = 10

    @staticmethod
    def _pos(coeff, t):
        return np.around(np.array([1, t, t ** 2]) @ coeff.T)                                 # :121

    @staticmethod
    def _vel(coeff, t):
        return np.array([1, 2 * t]) @ coeff[:, 1:].T                                         # :122

    def plan(self, drone, update_t):
        if len(self.trajectory) != 0:
            return True
        tgt = self.target[:2]
        self.trajectory = Trajectory2D()
        start = _Node(np.array([drone.x, drone.y]), drone.velocity, 0, tgt, -1, None, 0)
        open_set, closed_set = {start.index: start}, {}
        grid, trackers = drone.map, drone.trackers
        H = self.dt
        goal, itr = None, 0
        while True:
            itr += 1
            if len(open_set) == 0 or itr >= 100:                                             # :149
                break
            cid = min(open_set, key=lambda o: open_set[o].total_cost)
            cur = open_set[cid]
            if norm(cur.position - tgt) <= self.search_threshold:
                goal = cur
                break
            del open_set[cid]
            closed_set[cid] = cur
            succ = []
            px, py = cur.position[0], cur.position[1]
            vx, vy = cur.velocity[0], cur.velocity[1]
            for ax in self.u_space:
                for ay in self.u_space:
                    v_end = np.array([1, 2 * H]) @ np.array([[vx, vy], [ax / 2, ay / 2]])
                    if not (norm(v_end) < self.params.drone_max_speed):
                        continue
                    coeff = np.array([[px, vx, ax / 2], [py, vy, ay / 2]])
                    ok = True
                    for t in np.arange(0, H, H / self.sample_num):
                        if not self.is_free(self._pos(coeff, t), t + cur.itr * H, grid, trackers):
                            ok = False
                            break
                    if ok:
                        p_end = np.around(np.array([1, H, H ** 2]) @ np.array([[px, py], [vx, vy], [ax / 2, ay / 2]]))
                        succ.append(_Node(p_end, v_end, cur.cost + (ax ** 2 + ay ** 2) / 100 + 10, tgt,
                                          cur.index, coeff, cur.itr + 1))
            for n in succ:
                if n.index in closed_set:
                    continue
                if n.index not in open_set or open_set[n.index].cost > n.cost:
                    open_set[n.index] = n
        if goal is None:
            return False
        node = goal
        ts = np.arange(H, 0, -update_t)
        while node is not start:
            self.trajectory.positions.extend([self._pos(node.coeff, t) for t in ts])
            self.trajectory.velocities.extend([self._vel(node.coeff, t) for t in ts])
            self.trajectory.accelerations.extend([np.array([0, 0]) for _ in ts])
            node = closed_set[node.parent_index]
        self.trajectory.positions.reverse()
        self.trajectory.velocities.reverse()
        return True

    def replan_check(self, drone):
        """traj_planner.py:220-233 (swep_map is uint8, so i * dt truncates)."""
        occ = drone.map.grid_map
        swep = np.zeros_like(occ)
        s, dt = self.params.map_scale, self.params.dt
        for i, pos in enumerate(self.trajectory.positions):
            swep[int(pos[0] // s), int(pos[1] // s)] = i * dt
            for tr in drone.trackers:
                if tr.active and norm(tr.estimate_pos(i * dt) - pos) <= self.params.drone_radius + tr.radius:
                    self.trajectory.clear()
                    return True, swep
        if np.sum(np.where(occ == 1, 1, 0) * swep) > 0:
            self.trajectory.clear()
            return True, swep
        return False, swep


def _unavailable(name, why):
    class _U(Planner):
        def __init__(self, drone, params):
            raise NotImplementedError(f'planner {name!r}: {why}')
    _U.__name__ = name
    return _U


planner_list = {
    'Primitive': Primitive,
    'NoMove': NoMove,
    'MPC': _unavailable('MPC', 'needs the proprietary FORCESPRO solver (unusable in the reference too, '
                               'traj_planner.py:14-16,243)'),
    'Jerk_Primitive': _unavailable('Jerk_Primitive', 'not part of the accelerated hot path; register your own class '
                                                     'with planners.register_planner'),
}


def register_planner(name, cls):
    """Plug a planner class with the reference interface (e.g. the reference's own traj_planner classes)."""
    planner_list[name] = cls
