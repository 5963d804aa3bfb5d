import sys, torch
sys.path.insert(0, '.')
from benchpush_amd.envs.ship_ice import BatchedShipIceEnv, default_trials
E=4096
trials = default_trials(0.5, 100, base_seed=0)
env = BatchedShipIceEnv(E, cfg={"concentration": 0.5}, trials=trials)
env.reset()
g = torch.Generator(device=env.device); g.manual_seed(7)
age = torch.zeros(E, dtype=torch.int64, device=env.device); done=0
for t in range(300):
    a = (torch.rand(E, generator=g, device=env.device, dtype=torch.float64)*2-1)*0.5
    _, rew, term, _, info = env.step(a)
    age += 1; m = term.bool() | (age >= 300); done += int(m.sum()); age[m] = 0; env.reset(m)
    if t % 50 == 49: env.check_errors(); print(t, done, flush=True)
env.check_errors(); print("ok", done)
