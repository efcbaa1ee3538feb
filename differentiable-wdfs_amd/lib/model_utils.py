"""Drop-in for wdf_py/lib/model_utils.py: model weights <-> the JSON the C++ plugin embeds.

Schema (clipper_pot.py:298-331 writer, layers.py:51-70 reader, plugin loader
DiodePairNeuralModel.h:55-61):
    {"in_shape": [null, 2],
     "layers": [{"type": "dense", "shape": [null, out], "weights": [kernel[in][out], bias[out]],
                 "activation": "tanh" | "relu" | ""}, ...]}
Files written by the Keras pre-training carry a leading {"type": "unknown"} InputLayer entry
(model_utils.py:59-66) which readers skip (layers.py:57).
The reference's save_model_json walks a Keras model; here the models are
layers.DenseRootModel objects (the only kind the WDF path trains), so save_model_json takes
one of those (or anything with the same `.layers` list).
"""
import json
from json import JSONEncoder

import numpy as np

from wdf_hip import compat_tf as tf


class NumpyArrayEncoder(JSONEncoder):                      # model_utils.py:10-14
    def default(self, obj):
        if isinstance(obj, np.ndarray):
            return obj.tolist()
        if isinstance(obj, (np.floating, np.integer)):
            return obj.item()
        return JSONEncoder.default(self, obj)


def save_model_json(model):
    """DenseRootModel -> dict in the schema above (clipper_pot.py:299-323)."""
    layers = []
    for layer in model.layers:
        if layer is tf.nn.tanh:
            layers[-1]["activation"] = "tanh"
            continue
        if layer is tf.nn.relu:
            layers[-1]["activation"] = "relu"
            continue
        if hasattr(layer, "kernel") and hasattr(layer, "bias"):
            layers.append({
                "type": "dense",
                "shape": (None, int(layer.bias.shape[-1])),
                "weights": [layer.kernel.numpy()[0], layer.bias.numpy()[0]],
                "activation": "",
            })
    return {"in_shape": (None, int(model.layers[0].kernel.shape[1])), "layers": layers}


def save_model(model, filename):
    with open(filename, "w") as outfile:
        json.dump(save_model_json(model), outfile, cls=NumpyArrayEncoder, indent=4)


def load_model_json(filename):
    with open(filename, "r") as f:
        return json.load(f)
