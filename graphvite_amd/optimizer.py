"""graphvite_amd.optimizer — the reference's optimizer classes (include/core/optimizer.h:44-319, bound at
include/bind.h:757-999) as plain Python objects; the arithmetic itself runs in the HIP kernels."""
from .base import auto
from .kernels import OptimizerSpec


class LRSchedule(object):
    """
    LRSchedule(*args, **kwargs)
    Learning Rate Schedule: LRSchedule(type='constant') or LRSchedule(schedule_function).

    Parameters:
        type (str, optional): 'linear' or 'constant'
        schedule_function (callable): function (batch_id, num_batch) -> multiplicative factor
    """

    def __init__(self, type="constant"):
        if isinstance(type, LRSchedule):
            self.type, self.schedule_function = type.type, type.schedule_function
        elif callable(type):
            self.type, self.schedule_function = "custom", type
        elif type == "linear":
            self.type, self.schedule_function = "linear", LRSchedule.linear_schedule
        elif type == "constant":
            self.type, self.schedule_function = "constant", LRSchedule.constant_schedule
        else:
            raise ValueError("Invalid schedule `%s`" % (type,))

    def __call__(self, batch_id, num_batch):
        return self.schedule_function(batch_id, num_batch)

    @staticmethod
    def linear_schedule(batch_id, num_batch):  # optimizer.h:77-79
        return max(1 - float(batch_id) / num_batch, 1e-4)

    @staticmethod
    def constant_schedule(batch_id, num_batch):
        return 1

    def info(self):
        return "lr schedule: %s" % self.type

    __repr__ = info


class _OptimizerMeta(type):
    """`Optimizer(...)` is a factory (python/graphvite/optimizer.py:30-46 + the implicit conversions of
    include/bind.h:793-794); the helper classes construct normally."""

    def __call__(cls, *args, **kwargs):
        if cls is not Optimizer:
            return super().__call__(*args, **kwargs)
        type = args[0] if args else kwargs.pop("type", auto)
        args = args[1:]
        if isinstance(type, Optimizer):
            return type
        if isinstance(type, str):
            if type not in _CLASSES:
                raise ValueError("Unknown optimizer `%s`" % type)
            return _CLASSES[type](*args, **kwargs)
        obj = cls.__new__(cls)
        if isinstance(type, float):
            obj._init("Default", 0, type, 0, "linear")
        elif isinstance(type, int) and type == auto:
            obj._init("Default", 0, 0.0, 0, "linear")
        else:
            raise ValueError("Only auto can be used for initializing a default optimizer. Please use a float "
                             "value if you want to specify the learning rate.")
        return obj


class Optimizer(metaclass=_OptimizerMeta):
    """
    Optimizer(type=auto, *args, **kwargs)
    Create an optimizer instance of any type ('SGD', 'Momentum', 'AdaGrad', 'RMSprop' or 'Adam'), or a
    default optimizer (`auto`, or a bare learning rate) that the solver resolves at build().
    """

    def _init(self, type, num_moment, lr, weight_decay, schedule):
        self.type, self.num_moment = type, num_moment
        self.init_lr = self.lr = float(lr)
        self.weight_decay = float(weight_decay)
        self.schedule = LRSchedule(schedule)
        self.momentum = self.alpha = self.beta1 = self.beta2 = self.epsilon = 0.0

    def apply_schedule(self, batch_id, num_batch):  # optimizer.h:132-134
        self.lr = self.init_lr * self.schedule(batch_id, num_batch)

    def spec(self):
        """Kernel-facing description (gvk_optimizer)."""
        hp0 = {"Momentum": self.momentum, "RMSprop": self.alpha, "Adam": self.beta1}.get(self.type, 0.0)
        return OptimizerSpec(self.type, self.init_lr, self.weight_decay, hp0, self.beta2, self.epsilon,
                             self.schedule.type)

    def info(self):
        lines = ["optimizer: %s" % self.type, "learning rate: %g, %s" % (self.init_lr, self.schedule.info()),
                 "weight decay: %g" % self.weight_decay]
        if self.type == "Momentum":
            lines.append("momentum: %g" % self.momentum)
        if self.type == "AdaGrad":
            lines.append("epsilon: %g" % self.epsilon)
        if self.type == "RMSprop":
            lines.append("alpha: %g, epsilon: %g" % (self.alpha, self.epsilon))
        if self.type == "Adam":
            lines.append("beta1: %g, beta2: %g, epsilon: %g" % (self.beta1, self.beta2, self.epsilon))
        return "\n".join(lines)

    __repr__ = info


class SGD(Optimizer):
    """SGD(lr=1e-4, weight_decay=0, schedule='linear')"""

    def __init__(self, lr=1e-4, weight_decay=0, schedule="linear"):
        self._init("SGD", 0, lr, weight_decay, schedule)


class Momentum(Optimizer):
    """Momentum(lr=1e-4, weight_decay=0, momentum=0.999, schedule='linear')"""

    def __init__(self, lr=1e-4, weight_decay=0, momentum=0.999, schedule="linear"):
        self._init("Momentum", 1, lr, weight_decay, schedule)
        self.momentum = float(momentum)


class AdaGrad(Optimizer):
    """AdaGrad(lr=1e-4, weight_decay=0, epsilon=1e-10, schedule='linear')"""

    def __init__(self, lr=1e-4, weight_decay=0, epsilon=1e-10, schedule="linear"):
        self._init("AdaGrad", 1, lr, weight_decay, schedule)
        self.epsilon = float(epsilon)


class RMSprop(Optimizer):
    """RMSprop(lr=1e-4, weight_decay=0, alpha=0.999, epsilon=1e-8, schedule='linear')"""

    def __init__(self, lr=1e-4, weight_decay=0, alpha=0.999, epsilon=1e-8, schedule="linear"):
        self._init("RMSprop", 1, lr, weight_decay, schedule)
        self.alpha, self.epsilon = float(alpha), float(epsilon)


class Adam(Optimizer):
    """Adam(lr=1e-4, weight_decay=0, beta1=0.999, beta2=0.99999, epsilon=1e-8, schedule='linear')"""

    def __init__(self, lr=1e-4, weight_decay=0, beta1=0.999, beta2=0.99999, epsilon=1e-8, schedule="linear"):
        self._init("Adam", 2, lr, weight_decay, schedule)
        self.beta1, self.beta2, self.epsilon = float(beta1), float(beta2), float(epsilon)


_CLASSES = {"SGD": SGD, "Momentum": Momentum, "AdaGrad": AdaGrad, "RMSprop": RMSprop, "Adam": Adam}

__all__ = ["Optimizer", "LRSchedule", "SGD", "Momentum", "AdaGrad", "RMSprop", "Adam"]
