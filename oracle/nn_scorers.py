"""CPU restatement (numpy) of the frame-at-a-time NN feature scorers and their helpers -- test infrastructure only.

  class_labels_init      Nn::ClassLabelWrapper::initMapping                 Nn/ClassLabelWrapper.cc:56-70
  class_label_scores     Nn::BatchFeatureScorer::getScore / FullFeatureScorer::calculateScore
                                                                            Nn/BatchFeatureScorer.cc:148-171, Nn/FeatureScorer.cc:186-231
  on_demand_scores       Nn::OnDemandFeatureScorer::calculateScore + LinearAndSoftmaxLayer::getScore
                                                                            Nn/FeatureScorer.cc:119-135, Nn/LinearAndActivationLayer.cc:154-160
  precomputed_scores     Nn::PrecomputedFeatureScorer::calculateScore       Nn/FeatureScorer.cc:291-310
  vector_xml             what Core::XmlWriter << Math::Vector<T> writes     Math/Vector.hh:357-367 (element names, size attribute)

Parity unpinned: the Nn TUs need cblas / Core::Configuration (boost) to build; the functions follow the cited lines and the
reference holds no vectors for them.  FLT_MAX = Core::Type<f32>::max.
"""
import numpy as np

FLT_MAX = np.float32(3.402823466e+38)


def class_labels_init(n_classes, disregard=()):
    mapping = np.full(n_classes, -1, np.int32)
    n_targets = 0
    for c in range(n_classes):
        if c not in disregard:
            mapping[c] = n_targets
            n_targets += 1
    return mapping, n_targets


def class_label_scores(net_scores, mapping):
    """net_scores [T, n_outputs] = -(network output) as the batch scorer holds it; result [T, n_classes]"""
    T = net_scores.shape[0]
    out = np.full((T, len(mapping)), FLT_MAX, np.float32)
    keep = mapping >= 0
    out[:, keep] = net_scores[:, mapping[keep]]
    return out


def on_demand_scores(act, W_out, bias_folded, frames, emissions, mapping=None, acc=np.float64):
    """act [T, H]; W_out [n_outputs, H]; bias_folded = bias - alpha * logPrior (removeLogPriorFromBias);
    score = -bias[o] - W[o] . act[frame]  (getScore: result = -bias; result -= dot)"""
    out = np.zeros(len(frames), np.float32)
    for p, (t, e) in enumerate(zip(frames, emissions)):
        o = e if mapping is None else mapping[e]
        if o < 0:
            out[p] = FLT_MAX
            continue
        dot = np.dot(W_out[o].astype(acc), act[t].astype(acc))
        out[p] = np.float32(np.float32(-bias_folded[o]) - np.float32(dot))
    return out


def precomputed_scores(x, log_prior, prior_scale, mapping=None):
    """score = -x[o]; score += scale * prior[o]  (two f32 roundings, no fma)"""
    n_classes = x.shape[1] if mapping is None else len(mapping)
    out = np.full((x.shape[0], n_classes), FLT_MAX, np.float32)
    for e in range(n_classes):
        o = e if mapping is None else mapping[e]
        if o < 0:
            continue
        pr = np.float32(np.float32(prior_scale) * log_prior[o])
        out[:, e] = (-x[:, o]).astype(np.float32) + pr
    return out


def vector_xml(values, type_name):
    """the document Core::XmlWriter produces for a Math::Vector (scientific notation for floats)"""
    if type_name == "f32":
        body = " ".join("%e" % float(v) for v in values)
    else:
        body = " ".join("%d" % int(v) for v in values)
    return '<?xml version="1.0" encoding="ISO-8859-1"?>\n<vector-%s size="%d">\n  %s \n</vector-%s>\n' % (type_name, len(values), body, type_name)
