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
    if (this.opts.editable) this._bind();
  }

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
    c.strokeStyle = '#9aa4b1'; c.lineWidth = 1;
    for (let i = 0; i < n; i++) for (let j = 0; j < n; j++) if (this.adj[i][j] && (i < j || !this.adj[j][i])) {
      c.beginPath(); c.moveTo(this.pos[i][0], this.pos[i][1]); c.lineTo(this.pos[j][0], this.pos[j][1]); c.stroke();
    }
    const r = n > 60 ? 4 : n > 25 ? 7 : 11;
    for (let i = 0; i < n; i++) {
      c.beginPath(); c.arc(this.pos[i][0], this.pos[i][1], r, 0, 2 * Math.PI);
      c.fillStyle = COLORS[this.roles[i]] || '#555';
      if (this.status.length && !this.status[i]) c.fillStyle = '#c9ccd1';
      c.fill();
      if (i === this.sel) { c.strokeStyle = '#000'; c.lineWidth = 2; c.stroke(); c.strokeStyle = '#9aa4b1'; c.lineWidth = 1; }
      if (i === this.start) { c.beginPath(); c.arc(this.pos[i][0], this.pos[i][1], r + 3, 0, 2 * Math.PI); c.strokeStyle = '#f39c12'; c.stroke(); c.strokeStyle = '#9aa4b1'; }
      if (n <= 60) { c.fillStyle = '#1c2530'; c.font = '10px sans-serif'; c.fillText(this.labels[i] !== undefined ? this.labels[i] : i, this.pos[i][0] + r + 2, this.pos[i][1] - r); }
    }
  };

  TopologyGraph.prototype._hit = function (ev) {
    const b = this.canvas.getBoundingClientRect(), x = (ev.clientX - b.left) * this.canvas.width / b.width, y = (ev.clientY - b.top) * this.canvas.height / b.height;
    let best = -1, bd = 15 * 15;
    for (let i = 0; i < this.n; i++) { const d = (this.pos[i][0] - x) ** 2 + (this.pos[i][1] - y) ** 2; if (d < bd) { bd = d; best = i; } }
    return [best, x, y];
  };

  TopologyGraph.prototype._bind = function () {
    const self = this; let moved = false;
    this.canvas.addEventListener('mousedown', ev => { const h = self._hit(ev); self.drag = h[0]; moved = false; });
    this.canvas.addEventListener('mousemove', ev => {
      if (self.drag < 0) return; const h = self._hit(ev); self.pos[self.drag] = [h[1], h[2]]; moved = true; self.draw();
    });
    window.addEventListener('mouseup', ev => {
      const d = self.drag; self.drag = -1;
      if (d < 0 || moved) return;
      if (self.sel < 0) self.sel = d;
      else if (self.sel === d) { self.sel = -1; if (self.opts.onSelect) self.opts.onSelect(d); }
      else {
        const v = self.adj[self.sel][d] ? 0 : 1; self.adj[self.sel][d] = v; self.adj[d][self.sel] = v; self.sel = -1;
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
