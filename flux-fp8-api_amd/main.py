"""`python main.py` -- launch the HTTP API, flag for flag the CLI of the reference's main.py (:7-148, :151-195).

Offload flags (-OF / -OA / -OT) and --compile are accepted for command-line compatibility and have no effect here: nothing is
offloaded with 288 GB of HBM and there is no torch.compile stage (the fused HIP kernels + hipGraph replace it).  The flow dtype is
bfloat16 (the reference CLI passes float16 to its fp16-accumulate cuBLAS path, which has no counterpart on this hardware).
"""
from __future__ import annotations

import argparse


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Launch the Flux API server (MI355X)")
    p.add_argument("-c", "--config-path", type=str, help="Path to the configuration file, if not provided, the model is loaded from command line arguments")
    p.add_argument("-p", "--port", type=int, default=8088, help="Port to run the server on")
    p.add_argument("-H", "--host", type=str, default="0.0.0.0", help="Host to run the server on")
    p.add_argument("-f", "--flow-model-path", type=str, help="Path to the flow model")
    p.add_argument("-t", "--text-enc-path", type=str, help="Path to the text encoder")
    p.add_argument("-a", "--autoencoder-path", type=str, help="Path to the autoencoder")
    p.add_argument("-m", "--model-version", type=str, choices=["flux-dev", "flux-schnell"], default="flux-dev")
    p.add_argument("-F", "--flux-device", type=str, default="cuda:0")
    p.add_argument("-T", "--text-enc-device", type=str, default="cuda:0")
    p.add_argument("-A", "--autoencoder-device", type=str, default="cuda:0")
    p.add_argument("-q", "--num-to-quant", type=int, default=20)
    p.add_argument("-C", "--compile", action="store_true", default=False, help="accepted, no effect (no torch.compile stage)")
    p.add_argument("-qT", "--quant-text-enc", type=str, default="qfloat8", choices=["qint4", "qfloat8", "qint2", "qint8", "bf16"], dest="quant_text_enc")
    p.add_argument("-qA", "--quant-ae", action="store_true", default=False, dest="quant_ae")
    p.add_argument("-OF", "--offload-flow", action="store_true", default=False, dest="offload_flow")
    p.add_argument("-OA", "--no-offload-ae", action="store_false", default=True, dest="offload_ae")
    p.add_argument("-OT", "--no-offload-text-enc", action="store_false", default=True, dest="offload_text_enc")
    p.add_argument("-PF", "--prequantized-flow", action="store_true", default=False, dest="prequantized_flow",
                   help="Load the flow model from a prequantized checkpoint (the state_dict of a calibrated model saved as safetensors)")
    p.add_argument("-nqfm", "--no-quantize-flow-modulation", action="store_false", default=True, dest="quantize_modulation")
    p.add_argument("-qfl", "--quantize-flow-embedder-layers", action="store_true", default=False, dest="quantize_flow_embedder_layers")
    return p.parse_args(argv)


def build_pipeline(args):
    from flux_pipeline import FluxPipeline  # lazy: `--help` returns without importing torch
    from util import ModelVersion, load_config

    if args.config_path:
        return FluxPipeline.load_pipeline_from_config_path(args.config_path, flow_model_path=args.flow_model_path)
    config = load_config(
        ModelVersion.flux_dev if args.model_version == "flux-dev" else ModelVersion.flux_schnell, flux_path=args.flow_model_path,
        flux_device=args.flux_device, ae_path=args.autoencoder_path, ae_device=args.autoencoder_device, text_enc_path=args.text_enc_path,
        text_enc_device=args.text_enc_device, flow_dtype="bfloat16", text_enc_dtype="bfloat16", ae_dtype="bfloat16",
        num_to_quant=args.num_to_quant, compile_extras=args.compile, compile_blocks=args.compile,
        quant_text_enc={"qfloat8": "float8", "bf16": None}.get(args.quant_text_enc, args.quant_text_enc), quant_ae=args.quant_ae,
        offload_flow=args.offload_flow, offload_ae=args.offload_ae, offload_text_enc=args.offload_text_enc,
        prequantized_flow=args.prequantized_flow, quantize_modulation=args.quantize_modulation,
        quantize_flow_embedder_layers=args.quantize_flow_embedder_layers)
    return FluxPipeline.load_pipeline_from_config(config)


def main(argv=None):
    import uvicorn

    from api import app

    args = parse_args(argv)
    app.state.model = build_pipeline(args)
    uvicorn.run(app, host=args.host, port=args.port)


if __name__ == "__main__":
    main()
