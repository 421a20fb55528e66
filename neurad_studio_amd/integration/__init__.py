"""Everything a neurad-studio checkout needs to run its hot path on libneurad_hip.so without editing the reference:

  * ``tinycudann/`` and ``nerfacc/`` -- import-name packages.  Put THIS directory on PYTHONPATH and
    ``import tinycudann`` (nerfstudio/utils/external.py:38-58) / ``import nerfacc`` (models/neurad.py:28,
    model_components/renderers.py:34) resolve to the HIP-backed shims (operator-level boundary).
  * ``neurad_hip`` -- the nerfstudio method plugin (``neurad-hip``): a subclass of the reference's own NeuRADModel
    whose fields / sampler / renderers are this package's, registered through
    ``NERFSTUDIO_METHOD_CONFIGS="neurad-hip=neurad_studio_amd.integration.neurad_hip:neurad_hip"`` or the
    ``nerfstudio.method_configs`` entry-point group (plugins/registry.py:34-79).
"""
