"""Generator loading for the inversion (mirror of spi/utils/load_utils.py:15-33).

``load_eg3d()`` returns a ``TriPlaneGenerator`` in eval mode with ``neural_rendering_resolution = 128``,
rebuilt from the checkpoint's ``init_args/init_kwargs`` and filled with its parameters and buffers --
what the reference does through ``legacy.load_network_pkl`` + ``misc.copy_params_and_buffers``.

Checkpoint ingestion never executes the module source embedded in EG3D pickles
(eg3d/torch_utils/persistence.py:120-128,181-204): a restricted unpickler maps
``_reconstruct_persistent_obj`` to a plain stub and only the tensors / constructor arguments are read.
With no checkpoint on disk (this environment), ``synthetic=True`` builds the ffhqrebalanced512-128
architecture with seeded random weights.
"""
import io
import os
import pickle
import torch

from ..configs import paths_config, hyperparameters
from ..training.triplane import TriPlaneGenerator, ffhq512_kwargs


class EasyDict(dict):
    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value


class PersistentStub:
    """What a persistent_class instance unpickles to here: its class name and raw state, nothing executed."""
    def __init__(self, meta):
        self.class_name = meta['class_name']
        self.state = meta['state']


class InertObject:
    """What a PLAIN (non-persistent) class of the reference's own namespaces unpickles to -- e.g. training.triplane.OSGDecoder,
    training.volumetric_rendering.renderer.ImportanceRenderer, ray_sampler.RaySampler, which EG3D pickles by module path.  The
    reference module is never imported: the object only keeps the state dict pickle hands it (NEWOBJ + BUILD)."""
    def __init__(self, *args, **kwargs):
        pass

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)
        else:
            self.__dict__['_pickle_state'] = state


_REF_NAMESPACES = ('training.', 'torch_utils.', 'dnnlib.', 'camera_utils', 'legacy')
_stub_classes = {}


def _inert_class(module, name):
    key = (module, name)
    if key not in _stub_classes:
        _stub_classes[key] = type(name, (InertObject,), {'__module__': 'spi_amd.utils.load_utils', '_ref_path': f'{module}.{name}'})
    return _stub_classes[key]


def _reconstruct_stub(meta):
    assert meta.get('type') == 'class'
    return PersistentStub(meta)


def _load_storage_from_bytes(b):
    """Stand-in for torch.storage._load_from_bytes, which a tensor's storage reduces to (`torch.save` of the storage in the legacy
    format, nested inside the network pickle).  The real function is `torch.load(BytesIO(b), weights_only=False)` -- a full,
    unrestricted unpickle of the nested blob, through which a crafted pickle could call anything.  Here the nested blob goes through
    torch's own weights-only unpickler (tensors / storages / plain containers only; anything else raises UnpicklingError)."""
    return torch.load(io.BytesIO(b), map_location='cpu', weights_only=True)


# EXACT (module, name) pairs a network pickle may reference: tensor / storage rebuilders, containers, numpy scalars and arrays,
# the plain torch.nn containers EG3D's persistent classes hold (OSGDecoder.net is a Sequential with a Softplus).  Anything else
# -- in particular dotted names such as ('torch', 'os.system'), which pickle protocol 4 resolves attribute by attribute --
# is refused.
_TORCH_DTYPES = ('float32', 'float64', 'float16', 'bfloat16', 'int64', 'int32', 'int16', 'int8', 'uint8', 'bool')
_STORAGES = ('FloatStorage', 'DoubleStorage', 'HalfStorage', 'BFloat16Storage', 'LongStorage', 'IntStorage', 'ShortStorage',
             'CharStorage', 'ByteStorage', 'BoolStorage', 'UntypedStorage')
_ALLOWED = frozenset(
    [('collections', 'OrderedDict'), ('_codecs', 'encode'),
     ('torch._utils', '_rebuild_tensor_v2'), ('torch._utils', '_rebuild_parameter'), ('torch._utils', '_rebuild_parameter_with_state'),
     ('torch._utils', '_rebuild_tensor'), ('torch.storage', '_load_from_bytes'), ('torch', 'Size'), ('torch', 'device'),
     ('torch.nn.parameter', 'Parameter'), ('torch._tensor', '_rebuild_from_type_v2'), ('torch', 'Tensor'),
     ('torch.nn.modules.container', 'Sequential'), ('torch.nn.modules.container', 'ModuleList'),
     ('torch.nn.modules.container', 'ModuleDict'), ('torch.nn.modules.activation', 'Softplus'),
     ('numpy', 'ndarray'), ('numpy', 'dtype'),
     ('numpy.core.multiarray', '_reconstruct'), ('numpy.core.multiarray', 'scalar'),
     ('numpy._core.multiarray', '_reconstruct'), ('numpy._core.multiarray', 'scalar')]
    + [('torch', n) for n in _TORCH_DTYPES + _STORAGES])
_BUILTINS = frozenset(('dict', 'list', 'tuple', 'set', 'frozenset', 'int', 'float', 'bool', 'str', 'bytes', 'slice', 'complex'))


class _RestrictedUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module == 'torch_utils.persistence' and name == '_reconstruct_persistent_obj':
            return _reconstruct_stub
        if module in ('dnnlib.util', 'dnnlib') and name == 'EasyDict':
            return EasyDict
        if module == 'torch.storage' and name == '_load_from_bytes':
            return _load_storage_from_bytes               # never the real one: see the shim's docstring
        if '.' not in name and ((module == 'builtins' and name in _BUILTINS) or (module, name) in _ALLOWED):
            return super().find_class(module, name)
        if '.' not in name and name.isidentifier() and (module + '.').startswith(_REF_NAMESPACES):
            return _inert_class(module, name)             # a class of the reference's code base: inert stand-in, nothing imported
        raise pickle.UnpicklingError(f'refusing to import {module}.{name} from a network pickle')


def _flatten(node, prefix, out):
    d = node.state if isinstance(node, PersistentStub) else node.__dict__
    skip = d.get('_non_persistent_buffers_set', set())
    for name, p in (d.get('_parameters') or {}).items():
        if p is not None:
            out[prefix + name] = p.detach()
    for name, b in (d.get('_buffers') or {}).items():
        if b is not None and name not in skip:
            out[prefix + name] = b.detach()
    for name, child in (d.get('_modules') or {}).items():
        if child is not None:
            _flatten(child, prefix + name + '.', out)


def read_network_pkl(f, key='G_ema'):
    """-> (init_args, init_kwargs, state_dict, extra attributes) of data[key] in an EG3D network pickle."""
    data = _RestrictedUnpickler(f).load()
    net = data[key]
    if not isinstance(net, PersistentStub):
        raise pickle.UnpicklingError(f'{key} is not a persistent_class object')
    sd = {}
    _flatten(net, '', sd)
    st = net.state
    extra = {k: st[k] for k in ('rendering_kwargs', 'neural_rendering_resolution') if k in st}
    return tuple(st.get('_init_args', ())), dict(st.get('_init_kwargs', {})), sd, extra


def _plain(o):
    if isinstance(o, dict):
        return {k: _plain(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return type(o)(_plain(v) for v in o)
    return o


def build_generator(init_args=(), init_kwargs=None, state_dict=None, device='cpu'):
    G = TriPlaneGenerator(*init_args, **_plain(init_kwargs)).eval().requires_grad_(False)
    if state_dict is not None:
        G.load_state_dict(state_dict, strict=True)
    return G.to(device)


def load_eg3d(reload_modules=True, device='cuda', network_pkl=None, synthetic=False, seed=0):
    if network_pkl is None:
        network_pkl = paths_config.EG3D_PATH
    if os.path.isfile(network_pkl):
        if network_pkl.endswith('.pt'):                       # this implementation's own {init_kwargs, state_dict} format
            blob = torch.load(network_pkl, map_location='cpu')
            G = build_generator((), blob['init_kwargs'], blob['state_dict'], device)
        else:
            with open(network_pkl, 'rb') as f:
                args, kwargs, sd, extra = read_network_pkl(f)
            G = build_generator(args, kwargs, sd, device)
            if 'rendering_kwargs' in extra:
                G.rendering_kwargs = _plain(extra['rendering_kwargs'])
    elif synthetic:
        torch.manual_seed(seed)
        G = build_generator((), ffhq512_kwargs(), None, device)
    else:
        raise FileNotFoundError(f'{network_pkl} not found (pass synthetic=True / --synthetic for a seeded random-init generator)')
    if hyperparameters.depth_resolution is not None:
        G.rendering_kwargs = dict(G.rendering_kwargs, depth_resolution=int(hyperparameters.depth_resolution))
    if hyperparameters.depth_resolution_importance is not None:
        G.rendering_kwargs = dict(G.rendering_kwargs, depth_resolution_importance=int(hyperparameters.depth_resolution_importance))
    G.neural_rendering_resolution = 128
    G.eval()
    return G


def load_bisenet(device=None, path=None):
    """spi/utils/load_utils.py:36-44: BiSeNet(19) with paths_config.BISENET_PATH loaded, eval mode, on the GPU."""
    from ..third_part.bisenet import BiSeNet
    from ..configs import global_config
    path = path or paths_config.BISENET_PATH
    if not os.path.isfile(path):
        raise FileNotFoundError(f'{path} not found (paths_config.BISENET_PATH: the face-parsing checkpoint `bisenet.pth`)')
    net = BiSeNet(19)
    net.load_state_dict(torch.load(path, map_location='cpu', weights_only=True))
    return net.to(device or global_config.device).eval()
