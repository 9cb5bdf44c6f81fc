"""TEST INFRASTRUCTURE (oracle): CPU restatement of torchcrepe.predict as the reference calls it for
f0_method='mangio-crepe' (src/vc_infer_pipeline.py:96-137: predict(audio, 16000, hop, 50, 1100, 'full',
batch_size=2*hop, pad=True), then NaN-gating and np.interp resampling to p_len).

torchcrepe 0.0.20 and librosa 0.9.1 are pip dependencies absent from /root/reference and from this image; the algorithm
below follows their published sources (torchcrepe/core.py, model.py, decode.py, convert.py; librosa/sequence.py
`viterbi`) from memory.  PARITY UNPINNED: there is no golden vector for this file; HIP-vs-oracle parity is what the
tests establish.  The dither torchcrepe adds in bins_to_frequency (scipy.stats.triang) is passed in explicitly."""
import numpy as np
import torch
import torch.nn.functional as F

PITCH_BINS, WINDOW = 360, 1024
BN_EPS = 0.0010000000474974513
CENTS_PER_BIN = 20


def frequency_to_bins(freq, ceil=False):
    cents = 1200.0 * np.log2(freq / 10.0)
    b = (cents - 1997.3794084376191) / CENTS_PER_BIN
    return int(np.ceil(b) if ceil else np.floor(b))


def model_forward(sd, frames):
    """Crepe.forward (embed=False): frames (B, 1024) -> (B, 360) sigmoid posteriors."""
    x = frames[:, None, :, None]
    n_layers = 6
    for i in range(n_layers):
        n = "conv%d" % (i + 1)
        w = sd[n + ".weight"]
        pad = (0, 0, 254, 254) if i == 0 else (0, 0, 31, 32)
        x = F.conv2d(F.pad(x, pad), w, sd[n + ".bias"], stride=(4, 1) if i == 0 else (1, 1))
        x = F.relu(x)
        x = F.batch_norm(x, sd[n + "_BN.running_mean"], sd[n + "_BN.running_var"], sd[n + "_BN.weight"], sd[n + "_BN.bias"],
                         False, 0.0, BN_EPS)
        x = F.max_pool2d(x, (2, 1), (2, 1))
    x = x.permute(0, 2, 1, 3).reshape(x.shape[0], -1)
    return torch.sigmoid(F.linear(x, sd["classifier.weight"], sd["classifier.bias"]))


def frames_of(audio, hop):
    """torchcrepe.preprocess with pad=True: (1, N) -> (total_frames, 1024) normalised frames."""
    total = 1 + audio.shape[1] // hop
    a = F.pad(audio, (WINDOW // 2, WINDOW // 2))
    fr = a[0].unfold(0, WINDOW, hop)[:total].clone()
    fr = fr - fr.mean(dim=1, keepdim=True)
    fr = fr / torch.max(torch.tensor(1e-10), fr.std(dim=1, keepdim=True))
    return fr


def viterbi_path(prob):
    """librosa.sequence.viterbi(prob (n_states, n_steps), transition) with torchcrepe's 12-bin triangular transition,
    uniform p_init, epsilon = tiny of prob's dtype (float32)."""
    n_states, n_steps = prob.shape
    xx, yy = np.meshgrid(range(PITCH_BINS), range(PITCH_BINS))
    transition = np.maximum(12 - abs(xx - yy), 0).astype(np.float64)
    transition = transition / transition.sum(axis=1, keepdims=True)
    eps = np.finfo(prob.dtype).tiny
    log_trans = np.log(transition + eps)
    log_prob = np.log(prob.T + eps)
    log_p_init = np.log(np.full(n_states, 1.0 / n_states) + eps)
    value = np.zeros((n_steps, n_states), dtype=np.float64)
    ptr = np.zeros((n_steps, n_states), dtype=np.int64)
    value[0] = log_prob[0] + log_p_init
    for t in range(1, n_steps):
        trans_out = value[t - 1] + log_trans.T
        ptr[t] = np.argmax(trans_out, axis=1)
        value[t] = log_prob[t] + trans_out[np.arange(n_states), ptr[t]]
    state = np.zeros(n_steps, dtype=np.int64)
    state[-1] = np.argmax(value[-1])
    for t in range(n_steps - 2, -1, -1):
        state[t] = ptr[t + 1, state[t + 1]]
    return state


def predict(sd, audio, hop, fmin=50.0, fmax=1100.0, batch_size=None, dither=None):
    """-> (pitch float32 (n_frames,), bins int64 (n_frames,), posteriors (n_frames, 360))."""
    audio = torch.as_tensor(audio).float().view(1, -1)
    fr = frames_of(audio, hop)
    total = fr.shape[0]
    batch_size = batch_size or total
    lo, hi = frequency_to_bins(fmin), frequency_to_bins(fmax, ceil=True)
    bins_all, post_all = [], []
    with torch.no_grad():
        for i in range(0, total, batch_size):
            p = model_forward(sd, fr[i:i + batch_size])                 # (b, 360)
            post_all.append(p)
            logits = p.t().clone()                                      # (360, b)
            logits[:lo] = -float("inf")
            logits[hi:] = -float("inf")
            probs = F.softmax(logits, dim=0).numpy()
            bins_all.append(viterbi_path(probs))
    bins = np.concatenate(bins_all)
    cents = torch.tensor(CENTS_PER_BIN * bins + 1997.3794084376191, dtype=torch.float32)
    if dither is not None:
        cents = cents + torch.as_tensor(dither, dtype=torch.float32)
    pitch = 10 * 2 ** (cents / 1200)
    return pitch.numpy(), bins, torch.cat(post_all).numpy()


def mangio_crepe_f0(sd, x, p_len, hop, dither=None):
    """VC.get_f0_crepe_computation (vc_infer_pipeline.py:96-137)."""
    x = x.astype(np.float32)
    x = x / np.quantile(np.abs(x), 0.999)
    pitch, bins, post = predict(sd, x, hop, 50.0, 1100.0, batch_size=hop * 2, dither=dither)
    p_len = p_len or x.shape[0] // hop
    source = np.array(pitch)
    source[source < 0.001] = np.nan
    target = np.interp(np.arange(0, len(source) * p_len, len(source)) / p_len, np.arange(0, len(source)), source)
    return np.nan_to_num(target), bins, post
