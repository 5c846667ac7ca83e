import sys, numpy as np
sys.path.insert(0, "/root/repo")
import rasr_amd
from oracle import OracleMfcc, MfccCfg
from tests import synth
ctx = rasr_amd.Context(0)
kw = {'nr_cepstrum_coefficients': 3, 'filter_width': 151.5338237512972, 'sample_rate': 44100.0, 'alpha': 0.95, 'length': 0.01, 'shift': 0.01, 'maximum_input_size': 0.01, 'spacing': 0.0, 'normalize': True, 'front_end': 'mfplp', 'nr_autocorrelation_coefficients': 25, 'type': 'trapeze', 'boundary': 'include-boundary', 'warping_function': 'mel'}
cfg = MfccCfg(44100.0, 0.01, 0.01, 0.95, 0.01, 1, kw['filter_width'], 0.0, 1, 3, 1, 1, 25, 0.33, 1, 1, 0)
o = OracleMfcc(cfg); fe = rasr_amd.MfccExtractor(ctx, **kw)
print("filters", o.n_filters, "frame_len", o.frame_len, "fft", o.fft_len)
rng = np.random.Generator(np.random.PCG64(5))
for trial in range(6):
    n = [66610, 2, 30000, 441, 5000, 100000][trial]
    for scale in (1.0, 0.01, 30.0):
        x = synth.waveform(n, seed=100 + trial) * np.float32(scale)
        y = fe.run(x); w = o.run(x); w2 = o.run(np.nextafter(x, np.float32(np.inf))); w3 = o.run(np.nextafter(x, np.float32(-np.inf)))
        fin = np.isfinite(w)
        err = np.abs(y - w)[fin]; bar = 1e-4 * np.abs(w[fin]) + 1e-4
        sens = np.maximum(np.abs(w2 - w), np.abs(w3 - w))[fin]
        k = int(np.argmax(err / bar))
        print("n %6d scale %5g frames %4d  worst err/bar %.2f (err %.3g, want %.4g)  oracle 1-ulp sensitivity there %.3g (x300 = %.3g)  max sens/bar %.3g" % (n, scale, w.shape[0], (err / bar).max(), err[k], w[fin][k], sens[k], 300 * sens[k], (sens / bar).max()))
