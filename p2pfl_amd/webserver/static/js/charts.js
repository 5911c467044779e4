// Tiny line-chart + geo scatter helpers (canvas), used by the statistics and monitoring pages.
(function (global) {
  const PALETTE = ['#d3322c', '#2c6fd3', '#1a7f37', '#8e44ad', '#f39c12', '#16a085', '#7f8c8d', '#c0392b', '#2980b9', '#27ae60'];

  // series: {label: [[step, value], ...]}
  function lineChart(canvas, series, title) {
    const c = canvas.getContext('2d'), W = canvas.width, H = canvas.height, L = 48, B = 22, T = 18, R = 8;
    c.clearRect(0, 0, W, H);
    let xmin = Infinity, xmax = -Infinity, ymin = Infinity, ymax = -Infinity;
    Object.values(series).forEach(p => p.forEach(q => { xmin = Math.min(xmin, q[0]); xmax = Math.max(xmax, q[0]); ymin = Math.min(ymin, q[1]); ymax = Math.max(ymax, q[1]); }));
    if (!isFinite(xmin)) { c.fillText('no data', L, H / 2); return; }
    if (xmax === xmin) xmax = xmin + 1;
    if (ymax === ymin) { ymax += 0.5; ymin -= 0.5; }
    const X = x => L + (x - xmin) / (xmax - xmin) * (W - L - R), Y = y => H - B - (y - ymin) / (ymax - ymin) * (H - B - T);
    c.strokeStyle = '#dde1e6'; c.fillStyle = '#55606e'; c.font = '10px sans-serif';
    for (let i = 0; i <= 4; i++) {
      const yv = ymin + (ymax - ymin) * i / 4; c.beginPath(); c.moveTo(L, Y(yv)); c.lineTo(W - R, Y(yv)); c.stroke();
      c.fillText(yv.toPrecision(3), 2, Y(yv) + 3);
      const xv = xmin + (xmax - xmin) * i / 4; c.fillText(Math.round(xv), X(xv) - 8, H - 6);
    }
    c.fillStyle = '#1c2530'; c.font = 'bold 11px sans-serif'; c.fillText(title || '', L, 12);
    Object.keys(series).forEach((k, i) => {
      const p = series[k]; c.strokeStyle = PALETTE[i % PALETTE.length]; c.lineWidth = 1.5; c.beginPath();
      p.forEach((q, j) => { if (j) c.lineTo(X(q[0]), Y(q[1])); else c.moveTo(X(q[0]), Y(q[1])); }); c.stroke();
      if (p.length === 1) { c.fillStyle = c.strokeStyle; c.fillRect(X(p[0][0]) - 2, Y(p[0][1]) - 2, 4, 4); }
    });
  }

  // nodes: [{latitude, longitude, status, idx, role}], links: [[i, j], ...] (equirectangular plot of the scenario's area)
  function geoMap(canvas, nodes, links, colors) {
    const c = canvas.getContext('2d'), W = canvas.width, H = canvas.height, P = 24;
    c.clearRect(0, 0, W, H); c.fillStyle = '#eef3f8'; c.fillRect(0, 0, W, H);
    const pts = nodes.filter(n => isFinite(parseFloat(n.latitude)) && isFinite(parseFloat(n.longitude)));
    if (!pts.length) { c.fillStyle = '#55606e'; c.fillText('no coordinates reported', P, H / 2); return; }
    let la0 = Infinity, la1 = -Infinity, lo0 = Infinity, lo1 = -Infinity;
    pts.forEach(n => { const la = parseFloat(n.latitude), lo = parseFloat(n.longitude); la0 = Math.min(la0, la); la1 = Math.max(la1, la); lo0 = Math.min(lo0, lo); lo1 = Math.max(lo1, lo); });
    if (la1 - la0 < 1e-6) { la1 += 0.5; la0 -= 0.5; } if (lo1 - lo0 < 1e-6) { lo1 += 0.5; lo0 -= 0.5; }
    const X = lo => P + (lo - lo0) / (lo1 - lo0) * (W - 2 * P), Y = la => H - P - (la - la0) / (la1 - la0) * (H - 2 * P);
    c.strokeStyle = '#b8c2cf'; c.strokeRect(P / 2, P / 2, W - P, H - P);
    c.fillStyle = '#55606e'; c.font = '10px sans-serif';
    c.fillText(la1.toFixed(2) + '°N', 2, P); c.fillText(la0.toFixed(2) + '°N', 2, H - P / 2); c.fillText(lo0.toFixed(2) + '°E', P, H - 4); c.fillText(lo1.toFixed(2) + '°E', W - 3 * P, H - 4);
    c.strokeStyle = '#9aa4b1';
    (links || []).forEach(l => { const a = nodes[l[0]], b = nodes[l[1]]; if (!a || !b) return;
      c.beginPath(); c.moveTo(X(parseFloat(a.longitude)), Y(parseFloat(a.latitude))); c.lineTo(X(parseFloat(b.longitude)), Y(parseFloat(b.latitude))); c.stroke(); });
    pts.forEach(n => { c.beginPath(); c.arc(X(parseFloat(n.longitude)), Y(parseFloat(n.latitude)), 6, 0, 2 * Math.PI);
      c.fillStyle = n.status ? ((colors || {})[n.role] || '#2c6fd3') : '#c9ccd1'; c.fill();
      c.fillStyle = '#1c2530'; c.fillText(n.idx, X(parseFloat(n.longitude)) + 8, Y(parseFloat(n.latitude)) - 6); });
  }

  global.P2Charts = {lineChart: lineChart, geoMap: geoMap, PALETTE: PALETTE};
})(window);
