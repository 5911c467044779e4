// Topology editor / viewer on a <canvas> (no third-party libraries: the target machines have no network).
// Reference UI: a 3d-force-graph editor that emits an adjacency matrix (fedstellar/webserver/templates/
// deployment.html:538-555); here a 2-D circular/force layout with click-to-link editing.
//
//   const g = new TopologyGraph(canvas, {editable: true, onChange: fn});
//   g.setMatrix(matrix, roles)      matrix: n x n 0/1 array, roles: array of role strings
//   g.matrix()                      current adjacency matrix
// Editing: click two nodes in a row to toggle the link between them; drag a node to move it.
(function (global) {
  const COLORS = {aggregator: '#d3322c', trainer: '#2c6fd3', server: '#1a7f37', proxy: '#8e44ad', idle: '#7f8c8d'};

  function TopologyGraph(canvas, opts) {
    this.canvas = canvas; this.ctx = canvas.getContext('2d'); this.opts = opts || {};
    this.n = 0; this.adj = []; this.roles = []; this.pos = []; this.labels = []; this.status = [];
    this.sel = -1; this.drag = -1; this.start = 0;
    // 3-D view (the reference designer is a 3d-force-graph): positions get a depth coordinate, the canvas shows an
    // orbit projection that can be rotated with the mouse (right button / shift-drag) or left spinning
    this.mode3d = false; this.yaw = 0.6; this.pitch = 0.35; this.z = []; this.linkDistance = 1.0; this._spin = null;
    if (this.opts.editable) this._bind();
  }

  TopologyGraph.prototype.set3d = function (on, spin) {
    this.mode3d = !!on;
    if (this.mode3d && this.z.length !== this.n) this._depth();
    if (this._spin) { clearInterval(this._spin); this._spin = null; }
    if (this.mode3d && spin) this._spin = setInterval(() => { this.yaw += 0.02; this.draw(); }, 50);
    this.draw();
  };

  // depth of a node: breadth-first layer from the start node, spread around the canvas plane
  TopologyGraph.prototype._depth = function () {
    const n = this.n, d = new Array(n).fill(-1), q = [Math.min(this.start, n - 1)];
    if (n) d[q[0]] = 0;
    while (q.length) { const i = q.shift(); for (let j = 0; j < n; j++) if ((this.adj[i][j] || this.adj[j][i]) && d[j] < 0) { d[j] = d[i] + 1; q.push(j); } }
    const mx = Math.max(1, ...d);
    this.z = d.map((v, i) => ((v < 0 ? mx : v) / mx - 0.5) * 260 + ((i * 37) % 41 - 20));
  };

  // canvas position of node i under the current view
  TopologyGraph.prototype._proj = function (i) {
    const w = this.canvas.width, h = this.canvas.height, s = this.linkDistance;
    let x = (this.pos[i][0] - w / 2) * s, y = (this.pos[i][1] - h / 2) * s;
    if (!this.mode3d) return [w / 2 + x, h / 2 + y, 1];
    let z = (this.z[i] || 0) * s;
    const cy = Math.cos(this.yaw), sy = Math.sin(this.yaw), cp = Math.cos(this.pitch), sp = Math.sin(this.pitch);
    const x1 = x * cy + z * sy, z1 = -x * sy + z * cy, y1 = y * cp - z1 * sp, z2 = y * sp + z1 * cp;
    const f = 700 / (700 + z2);
    return [w / 2 + x1 * f, h / 2 + y1 * f, f];
  };

  // grow / shrink the federation in place (the reference designer's "add node" / "remove node" menu entries)
  TopologyGraph.prototype.addNode = function (neighbour, role) {
    const n = this.n; this.adj.forEach(r => r.push(0)); this.adj.push(new Array(n + 1).fill(0));
    if (neighbour >= 0 && neighbour < n) { this.adj[n][neighbour] = 1; this.adj[neighbour][n] = 1; }
    const b = neighbour >= 0 && neighbour < n ? this.pos[neighbour] : [this.canvas.width / 2, this.canvas.height / 2];
    this.pos.push([Math.min(this.canvas.width - 20, b[0] + 45), Math.min(this.canvas.height - 20, b[1] + 30)]);
    this.roles.push(role || 'trainer'); this.n = n + 1; this.z = [];
    if (this.mode3d) this._depth();
    if (this.opts.onChange) this.opts.onChange(this.matrix());
    this.draw(); return n;
  };
  TopologyGraph.prototype.removeNode = function (i) {
    if (this.n <= 2 || i < 0 || i >= this.n) return false;
    this.adj.splice(i, 1); this.adj.forEach(r => r.splice(i, 1)); this.pos.splice(i, 1); this.roles.splice(i, 1);
    this.n -= 1; this.sel = -1; this.z = [];
    if (this.start === i) this.start = 0; else if (this.start > i) this.start -= 1;
    if (this.mode3d) this._depth();
    if (this.opts.onChange) this.opts.onChange(this.matrix());
    this.draw(); return true;
  };

  TopologyGraph.prototype.setMatrix = function (m, roles, labels, status) {
    const n = m.length, keep = (n === this.n);
    this.n = n; this.adj = m.map(r => r.slice()); this.roles = roles || []; this.labels = labels || []; this.status = status || [];
    if (!keep) {
      const w = this.canvas.width, h = this.canvas.height, R = Math.min(w, h) / 2 - 30;
      this.pos = [];
      for (let i = 0; i < n; i++) {
        const a = 2 * Math.PI * i / Math.max(n, 1) - Math.PI / 2;
        this.pos.push([w / 2 + R * Math.cos(a), h / 2 + R * Math.sin(a)]);
      }
      if (n > 2 && n <= 200 && !this.opts.circular) this._relax(60);
    }
    this.draw();
  };

  TopologyGraph.prototype.matrix = function () { return this.adj.map(r => r.slice()); };

  // a few iterations of a spring layout (Fruchterman-Reingold) starting from the circle
  TopologyGraph.prototype._relax = function (iters) {
    const n = this.n, w = this.canvas.width, h = this.canvas.height, k = Math.sqrt(w * h / Math.max(n, 1)) * 0.6;
    for (let it = 0; it < iters; it++) {
      const d = this.pos.map(() => [0, 0]), t = 12 * (1 - it / iters) + 1;
      for (let i = 0; i < n; i++) for (let j = i + 1; j < n; j++) {
        let dx = this.pos[i][0] - this.pos[j][0], dy = this.pos[i][1] - this.pos[j][1];
        const dist = Math.max(Math.hypot(dx, dy), 1), rep = k * k / dist;
        dx /= dist; dy /= dist;
        d[i][0] += dx * rep; d[i][1] += dy * rep; d[j][0] -= dx * rep; d[j][1] -= dy * rep;
        if (this.adj[i][j] || this.adj[j][i]) {
          const att = dist * dist / k;
          d[i][0] -= dx * att; d[i][1] -= dy * att; d[j][0] += dx * att; d[j][1] += dy * att;
        }
      }
      for (let i = 0; i < n; i++) {
        const m = Math.max(Math.hypot(d[i][0], d[i][1]), 1e-9), s = Math.min(m, t) / m;
        this.pos[i][0] = Math.min(w - 20, Math.max(20, this.pos[i][0] + d[i][0] * s));
        this.pos[i][1] = Math.min(h - 20, Math.max(20, this.pos[i][1] + d[i][1] * s));
      }
    }
  };

  TopologyGraph.prototype.draw = function () {
    const c = this.ctx, n = this.n;
    c.clearRect(0, 0, this.canvas.width, this.canvas.height);
    if (this.mode3d && this.z.length !== n) this._depth();
    const P = []; for (let i = 0; i < n; i++) P.push(this._proj(i));
    this._screen = P;
    c.strokeStyle = '#9aa4b1'; c.lineWidth = 1;
    for (let i = 0; i < n; i++) for (let j = 0; j < n; j++) if (this.adj[i][j] && (i < j || !this.adj[j][i])) {
      c.beginPath(); c.moveTo(P[i][0], P[i][1]); c.lineTo(P[j][0], P[j][1]); c.stroke();
      if (!this.adj[j][i]) { const mx = (P[i][0] + 2 * P[j][0]) / 3, my = (P[i][1] + 2 * P[j][1]) / 3; c.fillStyle = '#9aa4b1'; c.fillRect(mx - 2, my - 2, 4, 4); }   // one-way link
    }
    const r0 = n > 60 ? 4 : n > 25 ? 7 : 11;
    const order = [...Array(n).keys()].sort((a, b) => P[a][2] - P[b][2]);      // far nodes first
    order.forEach(i => {
      const r = r0 * (this.mode3d ? P[i][2] : 1);
      c.beginPath(); c.arc(P[i][0], P[i][1], r, 0, 2 * Math.PI);
      c.fillStyle = COLORS[this.roles[i]] || '#555';
      if (this.status.length && !this.status[i]) c.fillStyle = '#c9ccd1';
      c.fill();
      if (i === this.sel) { c.strokeStyle = '#000'; c.lineWidth = 2; c.stroke(); c.strokeStyle = '#9aa4b1'; c.lineWidth = 1; }
      if (i === this.start) { c.beginPath(); c.arc(P[i][0], P[i][1], r + 3, 0, 2 * Math.PI); c.strokeStyle = '#f39c12'; c.stroke(); c.strokeStyle = '#9aa4b1'; }
      if (n <= 60) { c.fillStyle = '#1c2530'; c.font = '10px sans-serif'; c.fillText(this.labels[i] !== undefined ? this.labels[i] : i, P[i][0] + r + 2, P[i][1] - r); }
    });
  };

  TopologyGraph.prototype._hit = function (ev) {
    const b = this.canvas.getBoundingClientRect(), x = (ev.clientX - b.left) * this.canvas.width / b.width, y = (ev.clientY - b.top) * this.canvas.height / b.height;
    let best = -1, bd = 15 * 15;
    const P = this._screen || [];
    for (let i = 0; i < this.n && i < P.length; i++) { const d = (P[i][0] - x) ** 2 + (P[i][1] - y) ** 2; if (d < bd) { bd = d; best = i; } }
    return [best, x, y];
  };

  TopologyGraph.prototype._bind = function () {
    const self = this; let moved = false, orbit = null;
    this.canvas.addEventListener('contextmenu', ev => ev.preventDefault());
    this.canvas.addEventListener('mousedown', ev => {
      if (self.mode3d && (ev.button === 2 || ev.shiftKey)) { orbit = [ev.clientX, ev.clientY]; return; }
      const h = self._hit(ev); self.drag = h[0]; moved = false; });
    this.canvas.addEventListener('mousemove', ev => {
      if (orbit) { self.yaw += (ev.clientX - orbit[0]) * 0.01; self.pitch += (ev.clientY - orbit[1]) * 0.01; orbit = [ev.clientX, ev.clientY]; self.draw(); return; }
      if (self.drag < 0 || self.mode3d) return;                       // nodes are moved in the flat view only
      const h = self._hit(ev), w = self.canvas.width, hh = self.canvas.height, sc = self.linkDistance;
      self.pos[self.drag] = [w / 2 + (h[1] - w / 2) / sc, hh / 2 + (h[2] - hh / 2) / sc]; moved = true; self.draw();
    });
    window.addEventListener('mouseup', ev => {
      if (orbit) { orbit = null; return; }
      const d = self.drag; self.drag = -1;
      if (d < 0 || moved) return;
      if (self.sel < 0) self.sel = d;
      else if (self.sel === d) { self.sel = -1; if (self.opts.onSelect) self.opts.onSelect(d); }
      else {
        const v = self.adj[self.sel][d] ? 0 : 1; self.adj[self.sel][d] = v;
        if (!ev.altKey) self.adj[d][self.sel] = v;                     // alt-click: a one-way link
        self.sel = -1;
        if (self.opts.onChange) self.opts.onChange(self.matrix());
      }
      self.draw();
    });
  };

  // adjacency generators mirroring utils/topologymanager.py (fully / ring / star / random k=3)
  TopologyGraph.generate = function (kind, n, seed) {
    const m = []; for (let i = 0; i < n; i++) m.push(new Array(n).fill(0));
    const link = (a, b) => { if (a !== b) { m[a][b] = 1; m[b][a] = 1; } };
    let s = seed || 1; const rnd = () => { s = (s * 1664525 + 1013904223) % 4294967296; return s / 4294967296; };
    if (kind === 'fully') { for (let i = 0; i < n; i++) for (let j = i + 1; j < n; j++) link(i, j); }
    else if (kind === 'star') { for (let i = 1; i < n; i++) link(0, i); }
    else { for (let i = 0; i < n; i++) link(i, (i + 1) % n);
      if (kind === 'random') for (let i = 0; i < n; i++) { link(i, (i + 2) % n); if (rnd() < 0.5) link(i, Math.floor(rnd() * n)); } }
    return m;
  };

  TopologyGraph.COLORS = COLORS;
  global.TopologyGraph = TopologyGraph;
})(window);
