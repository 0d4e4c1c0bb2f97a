"""TF1 variable naming for the estimator-style models, so that weights can be exchanged with the reference.

The only naming contract the reference itself relies on is FNN.warm_up (estimator/models/ranking/fnn.py:32-48): from a saved
FM it reads
    linear/linear_model/<key>_indicator/weights      [num_buckets, 1]      -> linear_variables[<key>]
    linear/linear_model/bias_weights                 [1]                   -> linear_variables["bias"]
    factorized/input_layer/<key>_embedding/embedding_weights  [num_buckets, D]  -> factorized_variables[<key>]
(the FM builds them under `tf.variable_scope("linear")` / `("factorized")`, estimator/models/feature_interaction/fm.py:43-52).
Dense layers follow tf.layers.dense's names: <scope>/dense[_i]/{kernel,bias}."""
import numpy as np
import torch


def export_slab(slab, linear_scope="linear", factorized_scope="factorized"):
    out = {}
    for k in slab.keys:
        out["%s/input_layer/%s_embedding/embedding_weights" % (factorized_scope, k)] = slab.embedding_weights(k).detach().cpu().numpy().copy()
        if slab.has_linear:
            out["%s/linear_model/%s_indicator/weights" % (linear_scope, k)] = slab.linear_weights(k).detach().cpu().numpy().reshape(-1, 1).copy()
    if slab.has_linear:
        out["%s/linear_model/bias_weights" % linear_scope] = slab.lin_bias.detach().cpu().numpy().copy()
    return out


def import_slab(slab, variables, linear_scope="linear", factorized_scope="factorized", strict=True):
    """Inverse of export_slab; names may carry TF's ':0' suffix.  Returns the list of names consumed."""
    v = {n[:-2] if n.endswith(":0") else n: a for n, a in variables.items()}
    used = []

    def take(name, dst, shape):
        if name not in v:
            if strict:
                raise KeyError("variable %r not found" % name)
            return
        a = np.asarray(v[name], dtype=np.float32).reshape(shape)
        dst.copy_(torch.from_numpy(a).to(dst.device))
        used.append(name)
    for k in slab.keys:
        take("%s/input_layer/%s_embedding/embedding_weights" % (factorized_scope, k), slab.embedding_weights(k),
             tuple(slab.embedding_weights(k).shape))
        if slab.has_linear:
            take("%s/linear_model/%s_indicator/weights" % (linear_scope, k), slab.linear_weights(k),
                 tuple(slab.linear_weights(k).shape))
    if slab.has_linear:
        take("%s/linear_model/bias_weights" % linear_scope, slab.lin_bias.data, (1,))
    return used


def export_store(store, scope_from="dnn", scope_to="dnn"):
    out = {}
    for key, p in store.vars.items():
        name = key.replace("__", "/")
        if name.startswith(scope_from + "/"):
            name = scope_to + name[len(scope_from):]
        out[name] = p.detach().cpu().numpy().copy()
    return out


def import_store(store, variables, scope_from="dnn", scope_to="dnn", strict=True):
    v = {n[:-2] if n.endswith(":0") else n: a for n, a in variables.items()}
    used = []
    for key, p in store.vars.items():
        name = key.replace("__", "/")
        if name.startswith(scope_to + "/"):
            name = scope_from + name[len(scope_to):]
        if name in v:
            p.data.copy_(torch.from_numpy(np.asarray(v[name], dtype=np.float32).reshape(tuple(p.shape))).to(p.device))
            used.append(name)
        elif strict:
            raise KeyError("variable %r not found" % name)
    return used


def warm_up_dicts(variables):
    """The two dicts FNN.warm_up builds from a saved FM (fnn.py:35-48), from a {name: array} mapping."""
    linear, factorized = {}, {}
    for name, a in variables.items():
        n = name[:-2] if name.endswith(":0") else name
        parts = n.split("/")
        if parts[0] == "linear":
            linear["bias" if "bias" in n else parts[2].replace("_indicator", "")] = np.asarray(a)
        elif parts[0] == "factorized":
            factorized[parts[2].replace("_embedding", "")] = np.asarray(a)
    return linear, factorized
