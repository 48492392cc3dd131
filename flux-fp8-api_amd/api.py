"""HTTP surface of the reference (api.py of aredden/flux-fp8-api) over the MI355X pipeline -- SURVEY.md §8(f) row 4.

Same two POST endpoints, request fields, defaults and status codes:
  /generate  GenerateArgs{prompt, width=720, height=1024, num_steps=24, guidance=3.5, seed=random in (0, MAX_RAND), strength=1.0,
             init_image=None (path or base64)}  ->  image/jpeg stream of FluxPipeline.generate(**args)          (reference api.py:54-86)
  /lora      LoraArgs{scale=1.0, path, name, action="load"|"unload"}  ->  {"status": "success"} | 400 invalid action | 500 with the
             exception text; unload uses `name` when given, else `path`                                         (reference api.py:89-122)
The app holds one pipeline in `app.state.model`; FastAPI runs these sync handlers on its threadpool, and the pipeline serialises
engine access with its own lock (modules/flux_model.py), so concurrent requests queue instead of interleaving.
"""
from __future__ import annotations

import random
from typing import Literal, Optional

from fastapi import FastAPI
from fastapi.responses import JSONResponse, StreamingResponse
from pydantic import BaseModel, Field

MAX_RAND = 2**32 - 1


class LoraArgs(BaseModel):
    scale: Optional[float] = 1.0
    path: Optional[str] = None
    name: Optional[str] = None
    action: Optional[Literal["load", "unload"]] = "load"


class LoraLoadResponse(BaseModel):
    status: Literal["success", "error"]
    message: Optional[str] = None


class GenerateArgs(BaseModel):
    prompt: str
    width: Optional[int] = Field(default=720)
    height: Optional[int] = Field(default=1024)
    num_steps: Optional[int] = Field(default=24)
    guidance: Optional[float] = Field(default=3.5)
    seed: Optional[int] = Field(default_factory=lambda: random.randint(1, MAX_RAND - 1), gt=0, lt=MAX_RAND)
    strength: Optional[float] = 1.0
    init_image: Optional[str] = None


app = FastAPI(title="fluxmi")


@app.post("/generate")
def generate(args: GenerateArgs):
    """JPEG bytes of one image; `init_image` + `strength` select img2img (flux_pipeline.py:399-420,459-523 of the reference)."""
    result = app.state.model.generate(**args.model_dump())
    return StreamingResponse(result, media_type="image/jpeg")


@app.post("/lora", response_model=LoraLoadResponse)
def lora_action(args: LoraArgs):
    """Fuse (`load`) or subtract-and-requantise (`unload`) a LoRA into the fp8 flow weights."""
    try:
        if args.action == "load":
            app.state.model.load_lora(args.path, args.scale, args.name)
        elif args.action == "unload":
            app.state.model.unload_lora(args.name if args.name else args.path)
        else:
            return JSONResponse(status_code=400, content={"status": "error", "message": f"Invalid action, expected 'load' or 'unload', got {args.action}"})
    except Exception as e:  # the reference reports every failure as a 500 with the message
        return JSONResponse(status_code=500, content={"status": "error", "message": str(e)})
    return JSONResponse(status_code=200, content={"status": "success"})
