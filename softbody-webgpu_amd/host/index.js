'use strict';
module.exports = Object.assign({},
    require('./engineMapping'), require('./engine'), require('./engineWorker'), require('./lock'),
    require('./scenes'), require('./native'), require('./render'), require('./halo'));
