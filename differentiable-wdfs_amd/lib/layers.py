"""Drop-in for wdf_py/lib/layers.py: DenseLayer and DenseRootModel (the tanh-MLP root).

Same names, constructor arguments, attributes and JSON loader semantics as the reference
(layers.py:7-82): kernel shape [1, in, out], bias [1, out] (:18-21), set_weights([kernel,
bias]) (:31-36), the JSON walk that skips non-dense entries and appends tf.nn.tanh /
tf.nn.relu after a layer according to its "activation" (:51-70), `.layers` holding the
DenseLayer objects and activation functions in order (clipper_pot.py:298-331 relies on it
to write the weights back to JSON), incident(x[B,1,2]) / reflected().

Inside a WDF circuit the network is not evaluated here: tf_wdf.Circuit (and the loop
recorder) lower `P1.incident(-model.reflected())` (clipper_pot.py:119-121) to the HIP
kernels of csrc/wdf_mlp.h, which evaluate it per sequence in registers.
"""
import numpy as np

from wdf_hip import compat_tf as tf


_ACTIVATIONS = {"tanh": tf.nn.tanh, "relu": tf.nn.relu}      # JSON "activation" -> entry appended to .layers


class DenseLayer(tf.Module):
    """Dense layer without weights sharing (layers.py:7-39): kernel [1, in, out], bias [1, out]."""

    def __init__(self, in_size, out_size, kernel_init=None, bias_init=None):
        super().__init__()
        kernel_init = kernel_init or tf.keras.initializers.Orthogonal()
        bias_init = bias_init or tf.keras.initializers.Zeros()
        self.kernel = tf.Variable(self.init_weights(in_size, out_size, kernel_init), dtype=tf.float32)
        self.bias = tf.Variable(self.init_bias(out_size, bias_init), dtype=tf.float32)

    def init_weights(self, size1, size2, initializer):
        return [initializer(shape=(size1, size2))]

    def init_bias(self, size, initializer):
        return [initializer(shape=(size,) if np.isscalar(size) else size)]

    def set_weights(self, json_weights):
        kernel, bias = json_weights[0], json_weights[1]
        self.kernel.assign(np.array([kernel]))
        self.bias.assign(np.array([bias]))

    def __call__(self, input):  # noqa: A002
        return tf.matmul(input, self.kernel) + self.bias


class DenseRootModel(tf.Module):
    '''Root WDF model made of dense layers (layers.py:42-82).'''

    def __init__(self, json, verbose=False):
        super().__init__()
        self.a = tf.Variable(initial_value=tf.zeros(1), name="incident_wave", trainable=False)
        self.b = tf.Variable(initial_value=tf.zeros(1), name="reflected_wave", trainable=False)
        self.layers = []
        width = json["in_shape"][-1]
        for entry in json["layers"]:
            if entry["type"] != "dense":                     # e.g. the Keras InputLayer written as "unknown"
                continue
            out_width = entry["shape"][-1]
            if verbose:
                print(f"Adding Dense layer with size [{width}, {out_width}]")
            dense = DenseLayer(width, out_width)
            dense.set_weights(entry["weights"])
            self.layers.append(dense)
            if entry["activation"] in _ACTIVATIONS:
                self.layers.append(_ACTIVATIONS[entry["activation"]])
            width = out_width

    def incident(self, x):
        self.a = x[:, :, 0]
        self.model_in = x

    def reflected(self):
        x = self.model_in
        if hasattr(x, "__wdf_root__"):
            self.b = x.__wdf_root__(self)        # recorded loop: becomes the kernel's root
            return self.b
        for layer in self.layers:
            x = layer(x)
        self.b = x
        return self.b
