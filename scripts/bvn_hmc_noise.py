"""BGM(use_bnn=True): HMC on the re-perturbed target (reference as written, 'fresh') vs one weight draw ('frozen') -- draw
variance, reconstruction and imputation error of a small model (DESIGN_HISTORY.md section 7b).
usage: python scripts/bvn_hmc_noise.py [fresh|frozen]"""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from tests.test_gpu_bgm_bnn import _params, _linear_panel
from bayesgm_amd.models import BGM
n, p, q = 640, 12, 4
data = _linear_panel(n, p, q)
model = BGM(_params("gpurun_out/bvn_hmc_noise", p, q, bnn_mcmc_noise=sys.argv[1] if len(sys.argv) > 1 else "fresh"), random_seed=7)
model.fit(data, batch_size=32, epochs=40, epochs_per_eval=40, use_egm_init=True, egm_n_iter=60, egm_batches_per_eval=60, verbose=0)
miss = data[:150].copy()
rs = np.random.RandomState(5)
miss[rs.uniform(size=miss.shape) < 0.2] = np.nan
obs = ~np.isnan(miss)
# reconstruction from the trained latents
xr, _ = model._decode(model.data_z[:150], False)
print("train-z recon mse all cells", np.mean((xr - data[:150]) ** 2), "missing cells", np.mean((xr - data[:150])[~obs] ** 2))
for burn, nm, step in ((300, 100, 0.05), (1500, 200, 0.05)):
    draws = model.tfp_mcmc_sampler(miss, n_mcmc=nm, burn_in=burn, step_size=step, num_leapfrog_steps=5, seed=11)
    zm = draws.mean(0)
    xm, _ = model._decode(zm, False)
    print(burn, "hmc mean-z recon: observed cells", np.mean((xm - data[:150])[obs] ** 2), "missing cells", np.mean((xm - data[:150])[~obs] ** 2),
          "z draw var", draws.var(0).mean(), "corr with trained z", np.corrcoef(zm.ravel(), model.data_z[:150].cpu().numpy().ravel())[0, 1])
imp, itv = model.predict(miss, alpha=0.1, bs=64, n_mcmc=100, burn_in=300, step_size=0.05, num_leapfrog_steps=5, seed=11)
print("predict: missing mse", np.mean((imp - data[:150])[~obs] ** 2), "baseline", np.mean((data[:150][~obs] - data[:150][~obs].mean()) ** 2))
imp2, _ = model.predict(miss, alpha=0.1, bs=1000, n_mcmc=100, burn_in=300, step_size=0.05, num_leapfrog_steps=5, seed=11)
print("predict bs=1000: missing mse", np.mean((imp2 - data[:150])[~obs] ** 2))
