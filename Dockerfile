# CPU image: per-process nodes over the TCP protocol, controller, web frontend (reference: Dockerfile:1-24).
FROM ubuntu:22.04

ENV TZ=Europe/Madrid \
    DEBIAN_FRONTEND=noninteractive \
    PYTHONUNBUFFERED=1

RUN apt-get update && apt-get install -y --no-install-recommends \
        python3 python3-dev python3-pip g++ git curl net-tools iproute2 && \
    rm -rf /var/lib/apt/lists/*

WORKDIR /p2pfl_amd
COPY requirements.txt .
RUN python3 -m pip install --no-cache-dir -r requirements.txt
COPY . .
# host runtime (AES-NI link cipher, PARAMS framer / reassembler, message-id ring)
RUN python3 -c "from p2pfl_amd.ops.build import build_host; build_host(verbose=True)"

ENTRYPOINT ["python3"]
CMD ["app/main.py", "--webserver"]
